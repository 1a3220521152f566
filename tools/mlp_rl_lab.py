#!/usr/bin/env python3
"""Block tail in one launch (tr_mlp_fused_resid_ln_bf16) against the sequence it replaces: Mlp pair -> residual add + LayerNorm.

    python tools/mlp_rl_lab.py [rows ...]

Checks: x_new against x + float(Mlp pair output) (differs by the bf16 rounding of the fc2 output that the fused form does not do), xn_next
against torch's LayerNorm of the kernel's own x_new (<= 1 bf16 ulp), stream-K == whole-block schedule bit for bit; then HIP-event times of
(fused Mlp + layernorm launch) vs the one launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tokenreduction_amd import ops  # noqa: E402

D, Hd = 384, 1536
rows = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [50432, 35328, 24832, 32896, 1000, 129, 77]
dev = torch.device("cuda")
torch.manual_seed(0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


W1 = (0.05 * torch.randn(Hd, D, device=dev)).to(torch.bfloat16)
W2 = (0.05 * torch.randn(D, Hd, device=dev)).to(torch.bfloat16)
b1 = 0.1 * torch.randn(Hd, device=dev)
b2 = 0.1 * torch.randn(D, device=dev)
g = 1.0 + 0.2 * torch.randn(D, device=dev)
bt = 0.1 * torch.randn(D, device=dev)
pk = ops.mlp_pack(W1, W2, b2)
bad = 0
ROOM = 4 * 2 * 64 * 4 + 16      # a -DTR_DIAG_STAMPS build writes its stamps BEHIND xn_next: every output buffer here has the room


def alloc_y(M):
    return torch.zeros(M * D + ROOM, dtype=torch.bfloat16, device=dev)[:M * D].view(M, D)


for M in rows:
    xn = torch.randn(M, D, device=dev).to(torch.bfloat16)
    x0 = 2.0 * torch.randn(M, D, device=dev)
    d = ops.mlp_fused(xn, pk, b1, out=alloc_y(M))                   # bf16 Mlp output (bit-identical to the GEMM pair)
    want_x = x0 + d.float()
    outs = []
    for sk in (True, False):
        x = x0.clone()
        y = ops.mlp_fused_resid_ln(xn, pk, b1, b2, x, g, bt, 1e-6, xn_next=alloc_y(M), streamk=sk)
        torch.cuda.synchronize()
        outs.append((x, y))
    x, y = outs[0]
    same_sched = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1].view(torch.int16), outs[1][1].view(torch.int16))
    ex = float((x - want_x).abs().max())                           # <= half a bf16 ulp of the Mlp output
    tol_x = float(d.float().abs().max()) * 2.0 ** -8
    ref_y = torch.nn.functional.layer_norm(x, (D,), g, bt, 1e-6)
    ey = float(((y.float() - ref_y).abs() / (ref_y.abs() + 1e-2)).max())
    ok = same_sched and ex <= tol_x and ey <= 2.0 ** -7
    bad += 0 if ok else 1

    xs = x0.clone()
    ys, dd = alloc_y(M), alloc_y(M)

    def seq():
        ops.mlp_fused(xn, pk, b1, out=dd)
        ops.layernorm(xs, g, bt, 1e-6, delta=dd)

    def one():
        ops.mlp_fused_resid_ln(xn, pk, b1, b2, xs, g, bt, 1e-6, xn_next=ys)

    ts, to = timed(seq), timed(one)
    print(f"M={M:6d}  ok={ok} (schedules identical {same_sched}; |x - ref| {ex:.3g} <= {tol_x:.3g}; xn rel err {ey:.3g})   "
          f"fused Mlp + layernorm {ts:7.1f} us   one launch {to:7.1f} us   ratio {ts / to:.2f}", flush=True)
    if "--stamps" in sys.argv and M >= 9 * 128:       # a -DTR_DIAG_STAMPS build: per-step cycle stamps of workgroup 8 (written behind xn_next)
        buf = torch.zeros(M * D + 4 * 2 * 64 * 4 + 16, dtype=torch.bfloat16, device=dev)
        yv = buf[:M * D].view(M, D)
        ops.mlp_fused_resid_ln(xn, pk, b1, b2, xs, g, bt, 1e-6, xn_next=yv)
        torch.cuda.synchronize()
        st = buf[M * D:M * D + 2 * 64 * 4 * 4].view(torch.int64).view(2, 64, 4).cpu()
        for role, name in ((0, "P"), (1, "C")):
            dd_ = st[role]
            live = [i for i in range(64) if int(dd_[i, 0]) != 0]
            print(f"    {name} step lengths:", " ".join(str(int(dd_[live[k + 1], 0] - dd_[live[k], 0])) for k in range(len(live) - 1)))
            print(f"    {name} last recorded step: phases", [int(dd_[live[-1], j + 1] - dd_[live[-1], j]) for j in range(3)])
        es = buf[M * D + 60 * 4 * 4:M * D + 62 * 4 * 4].view(torch.int64).cpu()
        if int(es[0]):
            print("    C epilogue of the last block: table copy", int(es[1] - es[0]), " sum pass", int(es[2] - es[1]), " variance pass", int(es[3] - es[2]),
                  " reduce + rsqrt", int(es[4] - es[3]), " normalise + stores", int(es[5] - es[4]), "cycles")
print("ALL OK" if bad == 0 else f"{bad} SHAPES FAIL")
sys.exit(0 if bad == 0 else 1)
