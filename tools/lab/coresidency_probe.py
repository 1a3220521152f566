"""Lab: does a memory-bound LayerNorm launch of one stream run BESIDE a persistent GEMM-class launch of another stream, or only in its tail?
Stream A: n GEMM (or fused Mlp) launches; stream B: n LayerNorm launches; wall time of both together against each alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tokenreduction_amd import ops
D = 384
g = torch.Generator().manual_seed(1)
M = 50432
xn = torch.randn(M, D, generator=g).bfloat16().cuda()
wq, bq = (0.05 * torch.randn(3 * D, D, generator=g)).bfloat16().cuda(), torch.zeros(3 * D).cuda()
oq = torch.empty(M, 3 * D, dtype=torch.bfloat16, device="cuda")
w1, w2 = (0.05 * torch.randn(4 * D, D, generator=g)).bfloat16().cuda(), (0.05 * torch.randn(D, 4 * D, generator=g)).bfloat16().cuda()
b1, b2 = torch.zeros(4 * D).cuda(), torch.zeros(D).cuda()
pk = ops.mlp_pack(w1, w2, b2)
om = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
x = (2 * torch.randn(M, D, generator=g)).cuda()
d1, d2 = torch.randn(M, D, generator=g).bfloat16().cuda(), torch.randn(M, D, generator=g).bfloat16().cuda()
ga, be = torch.ones(D).cuda(), torch.zeros(D).cuda()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
n = 40


def run(fa, fb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fa:
        with torch.cuda.stream(sa):
            for _ in range(n):
                fa()
    if fb:
        with torch.cuda.stream(sb):
            for _ in range(n):
                fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


gemm = lambda: ops.gemm(xn, wq, bq, ops.TR_EPI_BF16, out=oq)
mlp = lambda: ops.mlp_fused(xn, pk, b1, out=om)
ln = lambda: ops.layernorm2(x, ga, be, 1e-6, d1, d2)
for name, fa in (("gemm_bf16_pc (qkv)", gemm), ("mlp_fused_kernel", mlp)):
    for _ in range(2):
        run(fa, ln)
    a, b, both = run(fa, None), run(None, ln), run(fa, ln)
    print(f"{name}: alone {a:6.1f} us, LayerNorm alone {b:6.1f} us, both streams {both:6.1f} us per pair (sum {a + b:6.1f})", flush=True)
