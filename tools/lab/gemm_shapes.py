"""Lab: TFLOP/s of tr_gemm_bf16 per shape of the headline forward (DeiT-S Top-K kr 0.7, B=256), each timed alone."""
import sys
import torch
sys.path.insert(0, ".")
from tokenreduction_amd import ops

dev = "cuda"
res = []
for tokens in (197, 138, 97, 68):
    M = 256 * tokens
    for name, N, K, epi in (("qkv", 1152, 384, ops.TR_EPI_BF16), ("proj", 384, 384, ops.TR_EPI_BF16), ("fc1", 1536, 384, ops.TR_EPI_GELU_BF16),
                            ("fc2", 384, 1536, ops.TR_EPI_BF16)):
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.zeros(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(5):
            ops.gemm(a, w, b, epi, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.gemm(a, w, b, epi, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        tiles = ((M + 255) // 256) * ((N + 127) // 128)
        print(f"tokens {tokens:3d} {name:4s} M={M:6d} N={N:4d} K={K:4d}: {us:7.2f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  tiles {tiles:5d} = {tiles / 256:.2f} rounds")
