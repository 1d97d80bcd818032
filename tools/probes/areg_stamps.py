#!/usr/bin/env python3
"""Per-workgroup wall-clock timeline of gemm_bx_areg_kernel at one shape (GPU box; library built with -DBXA_STAMPS=1).
PFOTGN_LIB=.../libpfotgn_stamps_bxa.so python tools/probes/areg_stamps.py [M N K]"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pfotgnrec_amd import _lib
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (53760, 704, 172)
dev = "cuda:0"
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
lib = _lib.load()
nbytes = lib.pfo_gemm_bf16x3_workspace_bytes(N, K)
iws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
def run():
    _lib.call("pfo_gemm_bf16x3", A.data_ptr(), K, B.data_ptr(), K, 0, C.data_ptr(), N, None, M, N, K, 0, iws.data_ptr(), nbytes, _lib.stream_ptr())
for _ in range(5): run()
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
n_wg = min(4096, ((M + 127) // 128 + 7) // 8 * 8 * ((N + 175) // 176))
buf = (ctypes.c_uint64 * (n_wg * 8))()
assert raw.pfo_debug_bxa_stamps(buf, n_wg * 8) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(n_wg, 8).astype(np.int64)
live = s[:, 4] > 0
s = s[live]
t0 = s[:, 0].min()
us = (s[:, :5] - t0) / 100.0
hw = s[:, 7] & 0xffffffff; xcc = s[:, 7] >> 32
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
slot = xcc * 1000 + se * 100 + sh * 16 + cu
print("shape M=%d N=%d K=%d: %d workgroups stamped, kernel span %.1f us" % (M, N, K, len(s), us[:, 4].max()))
names = ["launch -> loop start (prologue)", "main loop", "epilogue issue", "store drain (vmcnt 0)"]
for k in range(4):
    d = us[:, k + 1] - us[:, k]
    print("  %-34s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us" % (names[k], d.mean(), *np.percentile(d, [10, 50, 90])))
T = (K + 31) // 32
parts = np.stack([s[:, 5] & 0xffffffff, s[:, 5] >> 32, s[:, 6] & 0xffffffff, s[:, 6] >> 32], 1) / float(T)
print("  per k-tile, wavefront 0, shader cycles (mean over workgroups): MFMA + LDS reads %.0f | wait for loads %.0f | row split %.0f | barrier %.0f  (sum %.0f = %.2f us at 2.4 GHz)" %
      (*parts.mean(0), parts.mean(0).sum(), parts.mean(0).sum() / 2400.0))
if os.environ.get("STAMPS2"):
    loop_cyc = s[:, 3].astype(np.float64); loop_us = us[:, 2] - us[:, 1]
    print("  loop: %.0f shader cycles in %.2f us -> %.2f GHz while the kernel runs; unaccounted per k-tile (load issue at the top of a step) %.0f cycles" %
          (loop_cyc.mean(), loop_us.mean(), loop_cyc.mean() / loop_us.mean() / 1e3, loop_cyc.mean() / T - parts.mean(0).sum()))
life = us[:, 4] - us[:, 0]
print("  %-34s mean %6.2f us; sum of lifetimes / span = %.1f workgroups resident on average (of %d CU ids seen)" %
      ("lifetime", life.mean(), life.sum() / us[:, 4].max(), len(np.unique(slot))))
# start-time histogram: rounds
st = np.sort(us[:, 0])
print("  start times (us), deciles:", np.round(np.percentile(st, np.arange(0, 101, 10)), 1))
# one CU's timeline
c0 = slot == slot[0]
o = np.argsort(us[c0, 0])
print("  one CU (%d workgroups):" % c0.sum())
for row in us[c0][o][:12]:
    print("    start %6.2f  loop %6.2f  loop end %6.2f  stores issued %6.2f  drained %6.2f" % tuple(row))
