#!/usr/bin/env python3
"""Per-workgroup timeline of the weight-gradient tile (gemm_tn_group_bx_kernel<1>, four wavefronts) at one shape (GPU box;
library built with -DBXA_STAMPS=2): PFOTGN_LIB=.../libpfotgn_stamps_bxa.so python tools/probes/tn_stamps.py [M N K]"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pfotgnrec_amd import _lib
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (704, 172, 53760)
ASTAT = os.environ.get("ASTAT_SHAPE") == "1"        # the A-stationary image kernel at the d ctx' shape instead (same stamp layout)
dev = "cuda:0"
A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); C = torch.zeros(M, N, device=dev)
ws = torch.empty(40_000_000, device=dev)
if ASTAT:
    M, N, K = 53760, 704, 172
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.zeros(M, N, device=dev)
    nbytes = _lib.load().pfo_gemm_bf16x3_workspace_bytes(N, K)
    iws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
def run():
    if ASTAT:
        _lib.call("pfo_gemm_bf16x3", A.data_ptr(), K, B.data_ptr(), K, 0, C.data_ptr(), N, None, M, N, K, 0, iws.data_ptr(), nbytes, _lib.stream_ptr())
        return
    _lib.call("pfo_gemm_f32", A.data_ptr(), M, 1, B.data_ptr(), N, 1, C.data_ptr(), N, None, M, N, K, 0, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
for _ in range(5): run()
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
n_wg = 1024
buf = (ctypes.c_uint64 * (n_wg * 8))()
assert raw.pfo_debug_bxa_stamps(buf, n_wg * 8) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(n_wg, 8).astype(np.int64)
s = s[s[:, 4] > 0]
t0 = s[:, 0].min()
us = (s[:, [0, 1, 2, 4]] - t0) / 100.0
T = (s[:, 3] & 0xffffffff).astype(np.float64)
print("shape M=%d N=%d K=%d: %d workgroups stamped, %0.f k-tiles each, kernel span %.1f us" % (M, N, K, len(s), T.mean(), us[:, 3].max()))
for k, name in enumerate(["launch -> loop start (first tile staged)", "main loop", "slab stores + drain"]):
    d = us[:, k + 1] - us[:, k]
    print("  %-42s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us" % (name, d.mean(), *np.percentile(d, [10, 50, 90])))
parts = np.stack([s[:, 5] & 0xffffffff, s[:, 5] >> 32, s[:, 6] & 0xffffffff, s[:, 6] >> 32, s[:, 3] >> 32], 1) / T[:, None]
m = parts.mean(0)
if ASTAT:
    print("  per slot (two column tiles), wavefront 0, shader cycles: wait for the DMA %.0f | barrier %.0f | stores + DMA issue %.0f | MFMAs + LDS reads %.0f  (sum %.0f)" % (*m[:4], m[:4].sum()))
else:
    print("  per k-tile, wavefront 0, shader cycles: issue loads of t+1 %.0f | MFMAs + LDS reads %.0f | maxima (waits for the loads) %.0f | barrier %.0f | split + LDS stores + barrier %.0f  (sum %.0f)" % (*m, m.sum()))
loop_us = us[:, 2] - us[:, 1]
print("  loop: %.0f shader cycles in %.2f us -> %.2f GHz" % (s[:, 7].mean(), loop_us.mean(), s[:, 7].mean() / loop_us.mean() / 1e3))
print("  start times (us), deciles:", np.round(np.percentile(np.sort(us[:, 0]), np.arange(0, 101, 10)), 1))
