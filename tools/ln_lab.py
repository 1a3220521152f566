#!/usr/bin/env python3
"""Dev tool: times tr_layernorm_bf16 (residual add + LayerNorm) alone on the DeiT-S / DeiT-B row counts.
   TOKENREDUCTION_HIP_LIB=<alt .so> python tools/ln_lab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenreduction_amd import ops

for M, D in [(50432, 384), (35328, 384), (17664, 384), (25216, 768)]:
    x = torch.randn(M, D, device="cuda")
    d = (torch.randn(M, D, device="cuda") * 0.1).bfloat16()
    g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        ops.layernorm(x, g, b, 1e-6, delta=d)
    ts = []
    for _ in range(20):
        big.zero_()                                    # evict x / delta from the 256 MiB Infinity Cache, as a whole block does
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.layernorm(x, g, b, 1e-6, delta=d); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    byts = M * D * (4 + 2 + 4 + 2)
    us = ts[len(ts) // 2]
    hot = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.layernorm(x, g, b, 1e-6, delta=d); e1.record(); torch.cuda.synchronize()
        hot.append(e0.elapsed_time(e1) * 1e3)
    hot.sort()
    print(f"M={M:6d} D={D}: cold {us:7.1f} us  {byts / us / 1e6:5.2f} TB/s | back-to-back {hot[10]:7.1f} us {byts / hot[10] / 1e6:5.2f} TB/s")
