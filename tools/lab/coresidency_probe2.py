"""Lab: ONE gemm_bf16_pc launch on stream A, then ONE LayerNorm launch on stream B: when does each end?  (events on both streams, relative to a
common start event).  If the LayerNorm ends before the GEMM, its waves were resident beside the GEMM's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tokenreduction_amd import ops
D = 384
g = torch.Generator().manual_seed(1)
M = 50432
xn = torch.randn(M, D, generator=g).bfloat16().cuda()
wq, bq = (0.05 * torch.randn(3 * D, D, generator=g)).bfloat16().cuda(), torch.zeros(3 * D).cuda()
oq = torch.empty(M, 3 * D, dtype=torch.bfloat16, device="cuda")
x = (2 * torch.randn(M, D, generator=g)).cuda()
d1, d2 = torch.randn(M, D, generator=g).bfloat16().cuda(), torch.randn(M, D, generator=g).bfloat16().cuda()
ga, be = torch.ones(D).cuda(), torch.zeros(D).cuda()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
gemm = lambda: ops.gemm(xn, wq, bq, ops.TR_EPI_BF16, out=oq)
ln = lambda: ops.layernorm2(x, ga, be, 1e-6, d1, d2)
for _ in range(3):
    gemm(); ln()
torch.cuda.synchronize()
res = []
for rep in range(12):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    torch.cuda.synchronize()
    ev[0].record()
    sa.wait_event(ev[0]); sb.wait_event(ev[0])
    with torch.cuda.stream(sa):
        gemm(); ev[1].record(sa)
    with torch.cuda.stream(sb):
        ln(); ev[2].record(sb)
    torch.cuda.synchronize()
    res.append((ev[0].elapsed_time(ev[1]) * 1e3, ev[0].elapsed_time(ev[2]) * 1e3))
res.sort()
tag = os.path.basename(os.environ.get("TOKENREDUCTION_HIP_LIB", "product")) + " ln-block " + os.environ.get("TR_LN_BLOCK", "256")
print(tag, "| GEMM ends at / LayerNorm ends at (us after the common start), 12 runs:", " ".join(f"{a:.0f}/{b:.0f}" for a, b in res), flush=True)
