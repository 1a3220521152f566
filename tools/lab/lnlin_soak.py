"""Lab: back-to-back soak of tr_lnlin_bf16 (no host synchronisation between launches; fresh output buffers; the four stage shapes + two ragged
ones): every output and rewritten stream must equal LayerNorm + GEMM bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tokenreduction_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
D, N = 384, 1152
tot = 0
for M in (50432, 35328, 24832, 17408, 35331, 257 * 128, 1000):
    g0 = torch.Generator().manual_seed(M)
    x = (2.0 * torch.randn(M, D, generator=g0)).cuda()
    d1, d2 = torch.randn(M, D, generator=g0).bfloat16().cuda(), torch.randn(M, D, generator=g0).bfloat16().cuda()
    w, bias = (0.05 * torch.randn(N, D, generator=g0)).bfloat16().cuda(), (0.1 * torch.randn(N, generator=g0)).cuda()
    g, bt = (1.0 + 0.2 * torch.randn(D, generator=g0)).cuda(), (0.1 * torch.randn(D, generator=g0)).cuda()
    xr = x.clone()
    want = ops.gemm(ops.layernorm2(xr, g, bt, 1e-6, d1, d2), w, bias, ops.TR_EPI_BF16)
    pk = ops.lnlin_pack(w)
    res = []
    for it in range(n):
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        _, x_out = ops.lnlin(x, g, bt, 1e-6, pk, bias, d1=d1, d2=d2, out=out)
        res.append((out, x_out.clone() if x_out is not None else None))
        if len(res) >= 8:
            for o, xo in res:
                tot += 0 if (torch.equal(o.view(torch.int16), want.view(torch.int16)) and (xo is None or torch.equal(xo, xr))) else 1
            res = []
    for o, xo in res:
        tot += 0 if (torch.equal(o.view(torch.int16), want.view(torch.int16)) and (xo is None or torch.equal(xo, xr))) else 1
    print(f"M={M}: {n} launches, differing so far {tot}", flush=True)
print("ALL OK" if tot == 0 else f"{tot} DIFFER")
