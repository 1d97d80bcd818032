#!/usr/bin/env python3
"""A/B of the bf16x3 split contraction against the fp32 MFMA kernel (accuracy vs fp64 + time). GPU box only.
Run once per setting: PFO_GEMM_BF16X3=0|1 python tools/bench_gemm_bf16x3.py"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfotgnrec_amd import _lib

SHAPES = [(53760, 172, 704, "h1 nt"), (53760, 704, 172, "dctx' nt"), (12000, 876, 172, "QX nt"),
          (12000, 516, 520, "GRU ih nt"), (12000, 516, 172, "GRU hh nt"), (4099, 171, 44, "ragged")]
if os.environ.get("BX_SHAPES") == "old":
    SHAPES = [(53760, 348, 172, "QK' nt"), (53760, 172, 696, "h_pre nt"), (53760, 172, 172, "fc2 nt"),
              (53760, 172, 516, "fc1 nt"), (12000, 516, 520, "GRU ih nt"), (12000, 516, 172, "GRU hh nt"),
              (4099, 171, 44, "ragged")]
dev = "cuda:0"
torch.manual_seed(0)
ws = torch.empty(1 << 20, device=dev)
print("PFO_GEMM_BF16X3 =", os.environ.get("PFO_GEMM_BF16X3", "(default)"))
for M, N, K, label in SHAPES:
    # wide dynamic range per row to stress the split (exponents differ across k)
    A = torch.randn(M, K, device=dev) * torch.exp(2 * torch.randn(M, K, device=dev))
    B = torch.randn(N, K, device=dev) * torch.exp(2 * torch.randn(N, K, device=dev))
    C = torch.empty(M, N, device=dev)
    if os.environ.get("BX_API"):      # the pre-split-image path (includes the ~5 us image kernel of the small weight operand)
        nbytes = _lib.load().pfo_gemm_bf16x3_workspace_bytes(N, K)
        iws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        def run():
            _lib.call("pfo_gemm_bf16x3", A.data_ptr(), K, B.data_ptr(), K, 0, C.data_ptr(), N, None, M, N, K, 0,
                      iws.data_ptr(), nbytes, _lib.stream_ptr())
    else:
        def run():
            _lib.call("pfo_gemm_f32", A.data_ptr(), K, 0, B.data_ptr(), K, 0, C.data_ptr(), N, None, M, N, K, 0,
                      ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    rows = slice(0, min(M, 4096))
    ref = A[rows].double() @ B.double().T
    mag = A[rows].double().abs() @ B.double().abs().T          # sum |a||b|: the natural error scale
    err = ((C[rows].double() - ref).abs() / mag).max().item()
    print("%-10s M=%6d N=%4d K=%4d  %8.1f us  %6.1f TFLOP/s   max |err| / sum|a||b| = %.2e" %
          (label, M, N, K, us, 2.0 * M * N * K / us / 1e6, err))
