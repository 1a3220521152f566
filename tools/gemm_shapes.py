#!/usr/bin/env python3
"""Time every GEMM shape of the DeiT-S Top-K kr0.7 forward in isolation (HIP events, median of 20)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tokenreduction_amd import ops  # noqa: E402

B = 256
shapes = []
for n in (197, 138, 97, 68):
    M = B * n
    shapes += [("qkv", M, 1152, 384, ops.TR_EPI_BF16), ("proj", M, 384, 384, ops.TR_EPI_BF16),
               ("fc1", M, 1536, 384, ops.TR_EPI_GELU_BF16), ("fc2", M, 384, 1536, ops.TR_EPI_BF16)]
for name, M, N, K, epi in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(5):
        ops.gemm(a, w, b, epi, out=out)
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(a, w, b, epi, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    us = ts[len(ts) // 2]
    fl = 2.0 * M * N * K
    by = 2.0 * (M * K + N * K + M * N)
    print(f"{name:5s} M={M:6d} N={N:5d} K={K:5d}: {us:7.1f} us  {fl / us / 1e6:7.1f} TF/s  {by / us / 1e3:7.1f} GB/s  tiles {((M + 255) // 256) * ((N + 127) // 128)}")
