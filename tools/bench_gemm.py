#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA contraction at the shapes the C2 step launches (GPU box only)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfotgnrec_amd import _lib

SHAPES = [  # (M, N, K, a_km, b_km, label)
    (53760, 344, 172, 0, 0, "Q  nt"), (53760, 344, 344, 0, 0, "Wo nt"), (53760, 172, 516, 0, 0, "fc1 nt"),
    (53760, 172, 172, 0, 0, "fc2 nt"), (53760, 348, 172, 0, 1, "QK_h nn"), (53760, 344, 344, 0, 1, "dO nn"),
    (53760, 172, 348, 0, 0, "O_h nt"), (2560, 344, 344, 0, 0, "L2 Wo nt"), (2560, 172, 172, 0, 0, "L2 fc2"),
    (344, 344, 53760, 1, 1, "dWo tn"), (172, 348, 53760, 1, 1, "dWv_h tn"), (172, 172, 53760, 1, 1, "dW2 tn"),
    (516, 520, 12000, 1, 1, "dW_ih tn"), (12000, 516, 520, 0, 0, "GRU ih nt"),
]
dev = "cuda:0"
ws = torch.empty(30_000_000, device=dev)
for M, N, K, akm, bkm, label in SHAPES:
    A = torch.randn((K, M) if akm else (M, K), device=dev)
    B = torch.randn((K, N) if bkm else (N, K), device=dev)
    C = torch.empty((M, N), device=dev)
    def run():
        _lib.call("pfo_gemm_f32", A.data_ptr(), A.shape[1], akm, B.data_ptr(), B.shape[1], bkm, C.data_ptr(), N, None, M, N, K, 0,
                  ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("%-10s M=%6d N=%4d K=%6d  %8.1f us  %6.1f TFLOP/s" % (label, M, N, K, us, 2.0 * M * N * K / us / 1e6))
