#!/usr/bin/env python3
"""Time tr_sinkhorn in isolation: python3 tools/sinkhorn_lab.py [B N K iters]   (default: the first stage of sinkhorn_base at 384^2, B = 64).
TR_SINKHORN_REGS_OFF=1: the kernel that keeps the transport plan in global memory."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tokenreduction_amd import ops  # noqa: E402

a = [int(v) for v in sys.argv[1:]]
B, N, K, iters = (a + [64, 577, 144, 3][len(a):])[:4]
ldl = (K + 7) // 8 * 8
scores = (torch.randn(B, N, ldl, device="cuda") * 0.5).clamp(-1, 1)
for _ in range(3):
    ops.sinkhorn(scores, K, 1.0, iters)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    ops.sinkhorn(scores, K, 1.0, iters)
e1.record()
torch.cuda.synchronize()
print(f"B={B} N={N} K={K} iters={iters}: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us")
