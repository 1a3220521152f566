#!/usr/bin/env python3
"""TFLOP/s of the product GEMM per (rows, Linear) of a model's stages -- where tile quantisation on the small late stages costs time.

    python tools/gemm_shapes.py [D=768] [rows ...]      default rows: the stages of ats/dpcknn_base kr 0.5 at B = 128

Per shape: HIP-event time of 20 back-to-back launches of tr_gemm_bf16(TR_EPI_BF16) (qkv, proj, fc1, fc2 of a block with width D) and of
the four-Linear weight-gradient group (tr_linear_bwd_group through ops.linear_bwd_params per layer)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tokenreduction_amd import ops  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 768
rows = [int(a) for a in sys.argv[2:]] or [25216, 12672, 6400, 3200]
dev = torch.device("cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


tot = {}
for M in rows:
    line = []
    for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)):
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (0.02 * torch.randn(N, K, device=dev)).to(torch.bfloat16)
        bias = torch.zeros(N, device=dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        us = timed(lambda: ops.gemm(A, W, bias, ops.TR_EPI_BF16, out=out))
        dY = torch.randn(M, N, device=dev).to(torch.bfloat16)
        dW = torch.empty(N, K, device=dev)
        db = torch.empty(N, device=dev)
        usw = timed(lambda: ops.linear_bwd_params(dY, A, False, dW, db))
        fl = 2.0 * M * N * K
        line.append(f"{name} {us:6.1f} us {fl / us / 1e6:5.0f} TF | wgrad {usw:6.1f} us {fl / usw / 1e6:5.0f} TF")
        tot[M] = tot.get(M, 0.0) + 2 * us + usw
    print(f"M={M:6d}  " + "   ".join(line))
print("fwd + dgrad + wgrad per block, us:", {m: round(v, 1) for m, v in tot.items()})
