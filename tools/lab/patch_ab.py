"""A/B: fused patch embedding (tr_patch_embed_bf16) against im2col + GEMM + cls/pos, DeiT-S / DeiT-B shapes.  python tools/lab/patch_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tokenreduction_amd import ops

def bench(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for B, HW, D in ((256, 224, 384), (128, 224, 768), (64, 384, 768), (64, 224, 384), (32, 224, 384)):
    P = (HW // 16) ** 2
    img = torch.randn(B, 3, HW, HW, device="cuda")
    w = (torch.randn(D, 768, device="cuda") * 0.02).bfloat16()
    b = torch.randn(D, device="cuda") * 0.02
    cls, pos = torch.randn(D, device="cuda"), torch.randn(P + 1, D, device="cuda")
    x = torch.empty(B * (P + 1), D, device="cuda")
    def old():
        cols = ops.im2col(img, 16)
        ops.gemm(cols, w, b, ops.TR_EPI_PATCH_F32, out=x, aux=pos, aux_i=P)
        ops.cls_pos_rows(cls, pos, x, B, P + 1, D)
    new = lambda: ops.patch_embed(img, w, b, cls, pos)
    t_old, t_new = bench(old), bench(new)
    gb = (img.numel() * 4 + x.numel() * 4) / 1e9
    print(f"B={B} {HW}^2 D={D}: three launches {t_old:7.1f} us   fused {t_new:7.1f} us   ({gb / t_new * 1e6 / 1e3:.2f} TB/s algorithmic, "
          f"{2.0 * B * P * D * 768 / t_new / 1e6:.0f} TFLOP/s)")
