"""Which weight rows of a k-major image come out wrong? (debug probe, GPU box)"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_kernels import _gemm_bf16x3
rs = np.random.RandomState(300)
M, N, K = 300, 348, 520
A = rs.randn(M, K).astype(np.float64)
W = rs.randn(N, K).astype(np.float64)
A *= 2.0 ** rs.randint(-40, 40, size=(M, 1))
W *= 2.0 ** rs.randint(-20, 20, size=(N, 1))
ramp = 2.0 ** np.linspace(-12, 12, K)
A[0::7] *= ramp
A[1::7] *= ramp[::-1]
A[2::7, K // 2:] = 0
A[3::7] = 0
A[4::7] = 1e-42
W[5::11] = 0
W[6::11] *= ramp[::-1]
A = A.astype(np.float32); W = W.astype(np.float32)
ref = A.astype(np.float64) @ W.astype(np.float64).T
fa = np.maximum(np.abs(A).astype(np.float64), np.abs(A).max(1, keepdims=True).astype(np.float64) * 2.0 ** -18)
fw = np.maximum(np.abs(W).astype(np.float64), np.abs(W).max(1, keepdims=True).astype(np.float64) * 2.0 ** -18)
mag = fa @ fw.T + 1e-30
for km in (0, 1):
    got = _gemm_bf16x3(A, np.ascontiguousarray(W.T) if km else W, None, km).astype(np.float64)
    e2 = np.abs(got - ref) / mag
    print("  worst rows:", np.argsort(-e2.max(1))[:8], "row err", np.sort(e2.max(1))[-4:])
    err = e2.max(0)
    bad = np.nonzero(err > 4e-6)[0]
    print("k-major" if km else "row-major", "max err %.2e" % err.max(), "bad columns:", bad[:40], len(bad))
    if len(bad):
        print(" exponents of bad rows:", np.floor(np.log2(np.abs(W[bad]).max(1)))[:20])
        print(" ratio got/ref at bad:", (got[:, bad[:5]] / ref[:, bad[:5]])[0])
