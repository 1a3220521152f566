import os, sys
sys.path.insert(0, os.getcwd())
sys.argv = ["x"]
import torch
from tokenreduction_amd import ops
exec(open("tools/lab/lnlin_lab.py").read().split("shapes = [")[0])
for M, N in ((100864, 1152), (201728, 1152), (403456, 1152)):
    w = (0.05 * torch.randn(N, D, generator=g)).bfloat16().cuda()
    bias = (0.1 * torch.randn(N, generator=g)).cuda()
    x = (2 * torch.randn(M, D, generator=g)).cuda()
    d1, d2 = torch.randn(M, D, generator=g).bfloat16().cuda(), torch.randn(M, D, generator=g).bfloat16().cuda()
    pk = ops.lnlin_pack(w)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    xw = x.clone()
    xn = ops.layernorm2(xw, ga, be, 1e-6, d1, d2)
    t_two = ev_us(lambda: (ops.layernorm2(xw, ga, be, 1e-6, d1, d2), ops.gemm(xn, w, bias, ops.TR_EPI_BF16, out=out)), n=10)
    t_one = ev_us(lambda: ops.lnlin(x, ga, be, 1e-6, pk, bias, d1=d1, d2=d2, out=out), n=10)
    print(f"M={M} N={N}: both launches {t_two:7.1f} us | one launch {t_one:7.1f} us ({M // 128 / 256:.1f} blocks per workgroup)", flush=True)
    del x, d1, d2, out, xw, xn
