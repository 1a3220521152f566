"""Lab: TFLOP/s of the weight-gradient entry point (tr_linear_bwd_params: wgrad + bias sums + partial reduce) per shape."""
import sys
import torch
sys.path.insert(0, ".")
from tokenreduction_amd import ops

dev = "cuda"
for label, M, shapes in (("DeiT-S B=256 N=197", 256 * 197, ((1152, 384), (384, 384), (1536, 384), (384, 1536))),
                         ("DeiT-S B=256 N=97", 256 * 97, ((1152, 384), (384, 384), (1536, 384), (384, 1536))),
                         ("DeiT-B B=128 N=197", 128 * 197, ((2304, 768), (768, 768), (3072, 768), (768, 3072)))):
    for N, K in shapes:
        dy = torch.randn(M, N, device=dev).bfloat16()
        x = torch.randn(M, K, device=dev).bfloat16()
        dw = torch.empty(N, K, device=dev)
        db = torch.empty(N, device=dev)
        for _ in range(3):
            ops.linear_bwd_params(dy, x, dw=dw, db=db)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.linear_bwd_params(dy, x, dw=dw, db=db)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        # the vendor library on the same product (bf16 out, no bias sums): comparison only
        dwl = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            torch.mm(dy.t(), x, out=dwl)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            torch.mm(dy.t(), x, out=dwl)
        e1.record()
        torch.cuda.synchronize()
        usl = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{label}: dW[{N:4d},{K:4d}] M={M:6d}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s | library mm(dy.T, x) {usl:7.1f} us "
              f"{2.0 * M * N * K / usl / 1e6:6.1f} TFLOP/s")
