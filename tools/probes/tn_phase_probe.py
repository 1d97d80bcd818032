import sys, os, re
import torch
sys.path.insert(0, "/root/repo")
from pfotgnrec_amd import _lib
dev = "cuda:0"
torch.manual_seed(0)
M, N, K = 704, 172, 53760
WS = 40_000_000
ws = torch.zeros(WS, device=dev)
A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); C = torch.zeros(M, N, device=dev)
def run():
    _lib.call("pfo_gemm_f32", A.data_ptr(), M, 1, B.data_ptr(), N, 1, C.data_ptr(), N, None, M, N, K, 0, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
for _ in range(5): run()
torch.cuda.synchronize()
tiles, nsplit = 6, 85
off = WS - tiles * nsplit * 32 - 64
d = ws[off:off + tiles * nsplit * 32].view(-1, 8).cpu()
d = d[d[:, 6] > 0]
names = ["issue loads", "compute (LDS reads + MFMA)", "barrier 1", "split + LDS store (incl. wait for loads)", "barrier 2", "prologue"]
T = d[:, 6].mean().item()
print("waves", d.shape[0], "k-tiles per WG %.1f" % T)
tot = d[:, :5].sum(1).mean().item()
for q in range(5):
    print("%-42s %8.0f cycles/tile  %5.1f %%" % (names[q], d[:, q].mean().item() / T, 100 * d[:, q].mean().item() / tot))
print("loop total per tile %.0f (s_memtime ticks, 100 MHz => x%.0f for core cycles?)" % (tot / T, 1))
print("prologue", d[:, 5].mean().item())
