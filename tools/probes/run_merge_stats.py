import sys, numpy as np
sys.path.insert(0,'/root/repo')
from pfotgnrec_amd.synthetic import CONFIGS, make_graph
from pfotgnrec_amd.neighbor_finder import build_csr
cfg=CONFIGS['C2']; g=make_graph(cfg, with_prices=False); d=g.data
indptr,nbr,eidx,ts=build_csr(d.sources,d.destinations,d.edge_idxs,d.timestamps)
B=512;K=20
s=cfg.n_edges//2+200*B
rs=np.random.RandomState(0)
src=d.sources[s:s+B];dst=d.destinations[s:s+B];t=d.timestamps[s:s+B]
neg=rs.randint(cfg.n_users+1,cfg.n_users+cfg.n_items+1,size=B*3)
roots=np.concatenate([src,dst,neg]); rts=np.concatenate([t,t,np.repeat(t,3)])
def sample(nodes,tt):
    cnts=np.empty(len(nodes),np.int64); out=np.zeros((len(nodes),K),np.int64)
    for i,(v,x) in enumerate(zip(nodes,tt)):
        lo,hi=indptr[v],indptr[v+1]
        c=np.searchsorted(ts[lo:hi],x,side='left'); cnts[i]=c
        take=nbr[lo+max(0,c-K):lo+c]
        out[i,K-len(take):]=take
    return cnts,out
c2,n2=sample(roots,rts)
l1_nodes=np.concatenate([roots,n2.reshape(-1)]); l1_ts=np.concatenate([rts,np.repeat(rts,K)])
c1,n1=sample(l1_nodes,l1_ts)
keep=l1_nodes!=0
nodes=l1_nodes[keep]; cnt=c1[keep]
order=np.lexsort((np.arange(len(nodes)),cnt,nodes))
nodes=nodes[order];cnt=cnt[order]
M=len(nodes)
print("members",M,"distinct nodes",len(np.unique(nodes)), "items inst",(nodes>cfg.n_users).sum())
# per chunk of 4: rows flushed under (a) identical-list merging (b) shift merging
for RC in (4,8):
  rows_a=rows_b=0
  for c0 in range(0,M,RC):
    nd=nodes[c0:c0+RC]; cn=cnt[c0:c0+RC]
    # (a) runs = distinct (node,cnt) consecutive
    i=0
    while i<len(nd):
        j=i
        while j+1<len(nd) and nd[j+1]==nd[i] and cn[j+1]==cn[i]: j+=1
        rows_a+=min(cn[i],K)
        i=j+1
    # (b) groups: same node, delta<=64-K
    i=0
    while i<len(nd):
        j=i
        while j+1<len(nd) and nd[j+1]==nd[i] and cn[j+1]-cn[i]<=64-K: j+=1
        lo=max(0,cn[i]-K); hi=cn[j]
        rows_b+=hi-lo if cn[j]>0 else 0
        i=j+1
  tot=np.minimum(cnt,K).sum()
  print("RC",RC,"per-instance rows",tot,"identical-merge rows",rows_a,"shift-merge rows",rows_b)
it=nodes>cfg.n_users
print("item inst cnt spread per item: median delta between consecutive", np.median(np.diff(cnt[it])[np.diff(nodes[it])==0]))
u=~it
dn=np.diff(nodes[u])==0
print("user consecutive same-node pairs",dn.sum(),"of",u.sum()," with equal cnt",(np.diff(cnt[u])[dn]==0).sum())
