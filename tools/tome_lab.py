#!/usr/bin/env python3
"""Time tr_tome_match in isolation: python3 tools/tome_lab.py [N ...]   (B = 256, H = 6, r = 16; HIP events, 50 launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tokenreduction_amd import ops  # noqa: E402

B, H, r = 256, 6, 16
for N in [int(a) for a in sys.argv[1:]] or [197, 133, 69]:
    qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 1.5).bfloat16()
    for _ in range(5):
        ops.tome_match(qkv, B, N, H, min(r, (N - 1) // 2))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        ops.tome_match(qkv, B, N, H, min(r, (N - 1) // 2))
    e1.record()
    torch.cuda.synchronize()
    print(f"N={N:4d}: {e0.elapsed_time(e1) * 1e3 / 50:7.1f} us")
