"""Calibration only (never on the product path): what does the vendor library (hipBLASLt through torch.matmul) reach on the
hot GEMM shapes with the same random bf16 operands?  Puts the hand-written kernel's ~1.0 PFLOP/s of issued MFMA work in
context: is the gap to the 2.5 PFLOP/s datasheet peak the kernel's, or the chip's (power / DVFS) on this data?
usage (GPU box): python tools/blas_calib.py"""
import os
import sys

import torch

M = int(os.environ.get("GB_M", 122880))
SHAPES = [("qkv", 2304, 768), ("attout", 768, 768), ("ffn_up", 3072, 768), ("ffn_down", 768, 3072), ("square8k", 8192, 8192)]
dev = torch.device("cuda:0")
for fill in ("randn", "zeros"):
    for name, N, K in SHAPES:
        m = 8192 if name == "square8k" else M
        a = (torch.randn(m, K, device=dev) if fill == "randn" else torch.zeros(m, K, device=dev)).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.03 if fill == "randn" else torch.zeros(N, K, device=dev)).bfloat16()
        for _ in range(3):
            c = a @ w.t()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            c = a @ w.t()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("hipBLASLt bf16 %-6s %-9s M=%d N=%d K=%d  %7.3f ms  %6.0f TFLOP/s" % (fill, name, m, N, K, ms, 2.0 * m * N * K / ms / 1e9), flush=True)
