import sys; sys.path.insert(0,"."); sys.path.insert(0,"tests")
import numpy as np, torch
import pfotgnrec_amd as P
from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph
from oracle import tgn_oracle as T
from oracle.neighbor_finder import OracleNeighborFinder, build_adjacency
for (D,H,L,K,use_mem) in [(32,2,1,10,True),(172,2,2,8,True),(64,1,2,5,True),(24,4,3,3,True)]:
    torch.manual_seed(1234+D+H)
    cfg=SyntheticConfig("t",300,25,5000,D,L,K,H); g=make_graph(cfg,with_prices=False); d=g.data
    tgn=P.TGN(P.get_neighbor_finder(d,False),g.node_features,g.edge_features,"cuda:0",n_layers=L,n_heads=H,dropout=0.0,use_memory=use_mem,memory_dimension=D,message_function="identity")
    with torch.no_grad():
        tgn.time_encoder.w.bias.normal_(0,0.3)
        for att in tgn.embedding_module.attention_models:
            att.multi_head_target.in_proj_bias.normal_(0,0.1); att.multi_head_target.out_proj.bias.normal_(0,0.1)
    onf=OracleNeighborFinder(*build_adjacency(d.sources,d.destinations,d.edge_idxs,d.timestamps))
    names=[k for k in tgn.state_dict() if "layer_norm" not in k and not k.startswith("memory.")]
    ref=T.OracleTGN(onf,g.node_features,g.edge_features,{k:tgn.state_dict()[k].cpu().numpy() for k in names},L,H,use_mem)
    rs=np.random.RandomState(5); B=40; opt=P.FusedAdam(tgn,lr=1e-3)
    worst={}
    for step in range(4):
        s=2500+step*B
        sb,db,tb,eb=d.sources[s:s+B],d.destinations[s:s+B],d.timestamps[s:s+B],d.edge_idxs[s:s+B]
        neg=rs.randint(cfg.n_users+1,cfg.n_users+cfg.n_items+1,size=B*3)
        ref.P={k:tgn.state_dict()[k].detach().cpu().numpy().copy() for k in names}
        tgn.train(); opt.zero_grad()
        se,de,ne=tgn.compute_temporal_embeddings(sb,db,neg,tb,eb,K)
        rse,rde,rne=ref.compute_temporal_embeddings(sb,db,neg,tb,eb,K)
        emb=torch.cat([se,de,ne]); loss=P.bpr_loss(emb,B,3); loss.backward()
        rl,cache=T.bpr_loss(rse,rde.reshape(B,1,-1),rne.reshape(B,3,-1)); ds,dp,dn=T.bpr_loss_backward(cache)
        rg=ref.backward(np.concatenate([ds,dp.reshape(B,-1),dn.reshape(3*B,-1)]))
        for name,p in tgn.named_parameters():
            if name not in rg or np.abs(rg[name]).max()<1e-7: continue
            a=p.grad.cpu().numpy().astype(np.float64).ravel(); r=rg[name].astype(np.float64).ravel()
            mx=np.abs(a-r).max()/np.abs(r).max(); l2=np.linalg.norm(a-r)/np.linalg.norm(r)
            key=name.split(".")[-3:] ; key=".".join(key)
            w=worst.get(key,(0,0)); worst[key]=(max(w[0],mx),max(w[1],l2))
        opt.step()
    print("D%d H%d L%d:"%(D,H,L)," ".join("%s max %.1e l2 %.1e |"%(k,v[0],v[1]) for k,v in sorted(worst.items(),key=lambda kv:-kv[1][0])[:6]))
