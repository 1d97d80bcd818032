#!/usr/bin/env python3
"""Micro-benchmark of the weight-gradient form dW[M,N] = A[K,M]^T B[K,N] at the C2 step's shapes (GPU box only)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfotgnrec_amd import _lib

SHAPES = [(704, 172, 53760, "dWqk"), (172, 172, 53760, "dW2"), (516, 520, 12000, "dW_ih"), (704, 172, 2560, "L2 dWqk")]
dev = "cuda:0"
torch.manual_seed(0)
ws = torch.empty(40_000_000, device=dev)
for M, N, K, label in SHAPES:
    A = torch.randn(K, M, device=dev)
    B = torch.randn(K, N, device=dev)
    C = torch.zeros(M, N, device=dev)
    def run():
        _lib.call("pfo_gemm_f32", A.data_ptr(), M, 1, B.data_ptr(), N, 1, C.data_ptr(), N, None, M, N, K, 0,
                  ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ref = A[:4096].double().T @ B[:4096].double() if K > 4096 else A.double().T @ B.double()
    print("%-8s M=%4d N=%4d K=%6d  %8.1f us (GEMM + slab reduce)  %6.1f TFLOP/s" % (label, M, N, K, us, 2.0 * M * N * K / us / 1e6))
