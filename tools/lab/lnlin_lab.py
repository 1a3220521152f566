"""Lab: LayerNorm + Linear in one launch (tr_lnlin_bf16) against the LayerNorm launch + the GEMM launch, HIP-event us per call at the
headline's four stage shapes (norm1 + qkv: N = 1152; two pending residuals)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tokenreduction_amd import ops

D = 384
g = torch.Generator().manual_seed(1)
ga, be = (1 + 0.2 * torch.randn(D, generator=g)).cuda(), (0.1 * torch.randn(D, generator=g)).cuda()


def ev_us(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / n)
    return best


def dump_stamps(M, N):
    """-DTR_DIAG_STAMPS build: per-step stamps of one workgroup's MFMA wave 0 and LN wave 8 (cycles relative to the kernel's start)"""
    import numpy as np
    scr = next(iter(ops._LNLIN_SCRATCH.values()))
    G = (scr.numel() - 65536) // (2 * 128 * 384 * 2)
    st = scr[G * 2 * 128 * 384 * 2:].view(torch.int64).cpu().numpy().reshape(-1)[:2 * 64 * 4].reshape(2, 64, 4)
    t0 = int(st[0, 63, 0])
    print(f"  stamps M={M} N={N}: prologue (MFMA wave 0) start 0, LDS images {st[0,63,1]-t0}, first block normalised {st[0,63,2]-t0}, through barrier {st[0,63,3]-t0}")
    print(f"                     (LN wave 8)  {[int(v - t0) for v in st[1,63]]}")
    prev = int(st[0, 63, 3])
    for t in range(63):
        if st[0, t, 0] == 0:
            break
        m, l = st[0, t], st[1, t]
        print(f"  t={t:2d} MFMA: start +{int(m[0]-prev):6d} | windows 0-10 {int(m[1]-m[0]):5d} | vm/lgkm wait {int(m[2]-m[1]):5d} | barrier {int(m[3]-m[2]):5d}"
              f"   LN: work {int(l[1]-l[0]):5d} | drain {int(l[2]-l[1]):5d} | barrier {int(l[3]-l[2]):5d}")
        prev = int(m[3])


shapes = [(50432, 1152), (35328, 1152), (24832, 1152), (17408, 1152)]
if "--all" in sys.argv:
    shapes += [(50432, 384), (50432, 1536), (70001, 1152), (32768, 1152)]
for M, N in shapes:
    w = (0.05 * torch.randn(N, D, generator=g)).bfloat16().cuda()
    bias = (0.1 * torch.randn(N, generator=g)).cuda()
    x = (2 * torch.randn(M, D, generator=g)).cuda()
    d1, d2 = torch.randn(M, D, generator=g).bfloat16().cuda(), torch.randn(M, D, generator=g).bfloat16().cuda()
    pk = ops.lnlin_pack(w)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    xw = x.clone()
    xn = ops.layernorm2(xw, ga, be, 1e-6, d1, d2)
    t_ln = ev_us(lambda: ops.layernorm2(xw, ga, be, 1e-6, d1, d2))
    t_mm = ev_us(lambda: ops.gemm(xn, w, bias, ops.TR_EPI_BF16, out=out))
    t_two = ev_us(lambda: (ops.layernorm2(xw, ga, be, 1e-6, d1, d2), ops.gemm(xn, w, bias, ops.TR_EPI_BF16, out=out)))
    t_one = ev_us(lambda: ops.lnlin(x, ga, be, 1e-6, pk, bias, d1=d1, d2=d2, out=out))
    print(f"M={M:6d} N={N:5d}: layernorm2 {t_ln:6.1f} us, gemm {t_mm:6.1f} us ({2.0 * M * N * D / t_mm / 1e6:.0f} TFLOP/s), both {t_two:6.1f} us | "
          f"one launch {t_one:6.1f} us ({2.0 * M * N * D / t_one / 1e6:.0f} TFLOP/s incl. the norm)", flush=True)
    if "--stamps" in sys.argv:
        torch.cuda.synchronize()
        ops.lnlin(x, ga, be, 1e-6, pk, bias, d1=d1, d2=d2, out=out)
        torch.cuda.synchronize()
        dump_stamps(M, N)
