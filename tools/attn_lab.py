#!/usr/bin/env python3
"""Time the attention kernel in isolation: python3 tools/attn_lab.py [N ...]  (B=256, H=6, rocprof-free, HIP events).
TR_ATTN_LAB_SIZE=1: with a per-key size vector (ToMe's proportional attention / key masks: the kernel's key-bias path)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tokenreduction_amd import ops  # noqa: E402

B, H = 256, 6
for N in [int(a) for a in sys.argv[1:]] or [197, 138, 97, 68]:
    qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 1.5).bfloat16()
    size = torch.randint(1, 4, (B, N), device="cuda").float() if os.environ.get("TR_ATTN_LAB_SIZE") else None
    for _ in range(5):
        ops.attention(qkv, B, N, H, size=size)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        ops.attention(qkv, B, N, H, size=size)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    fl = 4.0 * B * H * N * N * 64
    print(f"N={N:4d}: {us:7.1f} us  {fl / us / 1e6:7.1f} TF/s  qkv+out {B * N * 4 * H * 64 * 2 / us / 1e3:7.1f} GB/s")
