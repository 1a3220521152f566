"""Lab: back-to-back soak of the three product forms of the fused Mlp launch -- plain / norm2 inside, stream-K / whole blocks -- on shapes whose
ranges end in short tails and on the stage shapes; fresh outputs, no host synchronisation in between; everything must equal the GEMM pair
(after a LayerNorm launch for the norm-inside form) bit for bit.  TR_MLP_FUSED_GRID=96 / 37 runs the hand-over chain in several rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tokenreduction_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
D, Hd = 384, 1536
tot = 0
for M in (257 * 128, 35328 + 3, 50432, 24832, 70001, 12800 + 5):
    g0 = torch.Generator().manual_seed(M)
    x = (2.0 * torch.randn(M, D, generator=g0)).cuda()
    dl = torch.randn(M, D, generator=g0).bfloat16().cuda()
    w1, w2 = (0.05 * torch.randn(Hd, D, generator=g0)).bfloat16().cuda(), (0.05 * torch.randn(D, Hd, generator=g0)).bfloat16().cuda()
    b1, b2 = (0.1 * torch.randn(Hd, generator=g0)).cuda(), (0.1 * torch.randn(D, generator=g0)).cuda()
    g, bt = (1.0 + 0.2 * torch.randn(D, generator=g0)).cuda(), (0.1 * torch.randn(D, generator=g0)).cuda()
    pk = ops.mlp_pack(w1, w2, b2)
    xn = ops.layernorm2(x, g, bt, 1e-6, dl, None, write_x=False)
    want = ops.gemm(ops.gemm(xn, w1, b1, ops.TR_EPI_GELU_BF16), w2, b2, ops.TR_EPI_BF16)
    bad = {}
    outs = []
    for it in range(n):
        for form in ("plain sk", "plain wb", "ln sk", "ln wb"):
            sk = form.endswith("sk")
            o = ops.mlp_fused(xn, pk, b1, streamk=sk) if form.startswith("plain") else ops.mlp_fused_ln(x, dl, g, bt, 1e-6, pk, b1, streamk=sk)
            outs.append((form, o))
        if len(outs) >= 16:
            for form, o in outs:
                if not torch.equal(o.view(torch.int16), want.view(torch.int16)):
                    bad[form] = bad.get(form, 0) + 1
            outs = []
    for form, o in outs:
        if not torch.equal(o.view(torch.int16), want.view(torch.int16)):
            bad[form] = bad.get(form, 0) + 1
    torch.cuda.synchronize()
    ops.mlp_fused_status()
    tot += sum(bad.values())
    print(f"grid {os.environ.get('TR_MLP_FUSED_GRID', 'CUs')} M={M}: {n} x 4 launches, differing: {bad}", flush=True)
print("ALL OK" if tot == 0 else f"{tot} DIFFER")
