"""Lab: the fused eval Mlp with its fc1 waves on v_mfma_f32_32x32x16_bf16 (library built with -DMF_P32: tools/lab/build_variant.sh mlp_p32 "-DMF_P32"
tr_mlp_fused.hip) -- accuracy against the GEMM pair and the float64 Mlp, HIP-event us per launch at the model's stage shapes.  Run once per
library (TOKENREDUCTION_HIP_LIB) and compare the lines."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle
from tokenreduction_amd import ops

D, Hd = 384, 1536
g = torch.Generator().manual_seed(1)
w1, w2 = (0.05 * torch.randn(Hd, D, generator=g)).bfloat16(), (0.05 * torch.randn(D, Hd, generator=g)).bfloat16()
b1, b2 = 0.1 * torch.randn(Hd, generator=g), 0.1 * torch.randn(D, generator=g)
w1d, w2d, b1d, b2d = w1.cuda(), w2.cuda(), b1.cuda(), b2.cuda()
pk = ops.mlp_pack(w1d, w2d, b2d)


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


print("library:", os.environ.get("TOKENREDUCTION_HIP_LIB", "product"))
for M in (300, 24832, 35328, 50432, 70001):
    x = torch.randn(M, D, generator=g).bfloat16()
    xd = x.cuda()
    pair = ops.gemm(ops.gemm(xd, w1d, b1d, ops.TR_EPI_GELU_BF16), w2d, b2d, ops.TR_EPI_BF16)
    out = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    got = ops.mlp_fused(xd, pk, b1d, out=out)
    ops.mlp_fused_status()
    ndiff = int((got.view(torch.int16) != pair.view(torch.int16)).sum())
    rows = torch.arange(0, M, max(1, M // 257))[:300]
    hid = oracle.gelu_erf(x[rows].double() @ w1.double().t() + b1.double()).float().bfloat16().double()
    ref = hid @ w2.double().t() + b2.double()
    e_f = ((got[rows.cuda()].cpu().double() - ref).norm() / ref.norm()).item()
    e_p = ((pair[rows.cuda()].cpu().double() - ref).norm() / ref.norm()).item()
    ulp = ((got.float() - pair.float()).abs() / (pair.float().abs() * 2.0 ** -7 + 1e-6)).max().item()
    t = ev_us(lambda: ops.mlp_fused(xd, pk, b1d, out=out))
    print(f"M={M:6d}: {t:7.1f} us ({4.0 * M * D * Hd / t / 1e6:.0f} TFLOP/s)   elements != pair {ndiff} of {got.numel()} (max {ulp:.2f} bf16 ulp)   "
          f"rel L2 vs float64 Mlp: fused {e_f:.3e}  pair {e_p:.3e}", flush=True)
