#!/usr/bin/env python3
"""Fused eval Mlp (tr_mlp_fused_bf16) against the two-GEMM pair it replaces: bit-equality and HIP-event time per launch.

    python tools/mlp_lab.py [rows ...]        default rows: the four stages of DeiT-S Top-K kr 0.7 at B = 256 + ragged / tiny cases

Per M: out_pair = gemm(gemm(x, W1, b1, GELU_BF16), W2, b2, BF16), out_fused = mlp_fused(x, pack(W1, W2, b2), b1); the two must be equal
bit for bit (same MFMA, same K order, same rounding points: csrc/tr_mlp_fused.hip)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tokenreduction_amd import ops  # noqa: E402

D, Hd = 384, 1536
STREAMK = "--no-streamk" not in sys.argv   # the hand-over scratch (stream-K schedule beyond 256 blocks); --no-streamk: whole blocks round-robin
STRESS = "--stress" in sys.argv            # re-run the fused launch 200 times per shape under uneven load and compare every output (race screen)
STAMPS = "--stamps" in sys.argv            # a -DTR_DIAG_STAMPS build (TOKENREDUCTION_HIP_LIB=...): print the per-step phase stamps of workgroup 8
rows = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [50432, 35328, 24832, 17408, 1000, 129, 128, 77, 1]
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
pk = ops.mlp_pack(W1, W2, b2)
bad = 0
for M in rows:
    x = torch.randn(M, D, device=dev).to(torch.bfloat16)
    h = torch.empty(M, Hd, dtype=torch.bfloat16, device=dev)
    o_pair = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    o_buf = torch.zeros(M * D + 4 * 2 * 64 * 4 + 16, dtype=torch.bfloat16, device=dev)          # + room for the diagnostic stamps
    o_fused = o_buf[:M * D].view(M, D)
    o_fused.fill_(float("nan"))

    def pair():
        ops.gemm(x, W1, b1, ops.TR_EPI_GELU_BF16, out=h)
        ops.gemm(h, W2, b2, ops.TR_EPI_BF16, out=o_pair)

    def fused():
        ops.mlp_fused(x, pk, b1, out=o_fused, streamk=STREAMK)

    pair()
    fused()
    torch.cuda.synchronize()
    same = torch.equal(o_pair.view(torch.int16), o_fused.view(torch.int16))
    ndiff = int((o_pair.view(torch.int16) != o_fused.view(torch.int16)).sum())
    maxd = float((o_pair.float() - o_fused.float()).abs().max())
    if STRESS:
        nbad = 0
        side = torch.cuda.Stream()
        junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
        for it in range(200):
            o_fused.fill_(float("nan"))
            if it % 3 == 0:                       # a copy kernel on a second stream: uneven load on the CUs' memory queues
                with torch.cuda.stream(side):
                    junk.add_(1)
            fused()
            torch.cuda.synchronize()
            if not torch.equal(o_pair.view(torch.int16), o_fused.view(torch.int16)):
                nbad += 1
                d = (o_pair.view(torch.int16) != o_fused.view(torch.int16))
                rows_bad = d.any(dim=1).nonzero().flatten()
                print(f"    stress iteration {it}: {int(d.sum())} elements differ, rows {rows_bad[:6].tolist()}..{int(rows_bad[-1])} ({len(rows_bad)} rows), cols {d.any(dim=0).nonzero().flatten()[:4].tolist()}..")
        print(f"M={M:6d}  stress: {nbad} of 200 launches differ", flush=True)
        bad += 1 if nbad else 0
        continue
    fl = 4.0 * M * D * Hd
    tp, tf = timed(pair), timed(fused)
    print(f"M={M:6d}  bit-identical={same} (differing {ndiff}, max |d| {maxd:.3g})   pair {tp:7.1f} us {fl / tp / 1e6:5.0f} TF   "
          f"fused {tf:7.1f} us {fl / tf / 1e6:5.0f} TF   ratio {tp / tf:.2f}", flush=True)
    bad += 0 if same else 1
    if STAMPS and M >= 9 * 128:
        o_buf.zero_()
        fused()
        torch.cuda.synchronize()
        ck = o_buf[M * D + 2 * 64 * 4 * 4:M * D + 2 * 64 * 4 * 4 + 8].view(torch.int64).cpu()
        if int(ck[1]) > 0:
            print(f"    in-kernel clock {int(ck[0]) / int(ck[1]) * 100:.0f} MHz over {int(ck[1]) / 100:.1f} us (workgroup 8's step loop)")
        st = o_buf[M * D:M * D + 2 * 64 * 4 * 4].view(torch.int64).view(2, 64, 4).cpu()
        for role, name in ((0, "P"), (1, "C")):
            d = st[role]
            rowsel = [i for i in range(64) if int(d[i, 0]) != 0][4:44]
            if not rowsel:
                continue
            body = sum(int(d[i, 1] - d[i, 0]) for i in rowsel) / len(rowsel)
            tail = sum(int(d[i, 2] - d[i, 1]) for i in rowsel) / len(rowsel)
            wait = sum(int(d[i, 3] - d[i, 2]) for i in rowsel) / len(rowsel)
            per = (int(d[rowsel[-1], 0]) - int(d[rowsel[0], 0])) / (len(rowsel) - 1)
            print(f"    stamps {name}: step {per:7.0f} cycles = MFMA phase {body:6.0f} + {'GELU + h write' if role == 0 else 'epilogue      '} {tail:6.0f} + wait/barrier {wait:6.0f}")
            if "--steps" in sys.argv:          # every recorded step of workgroup 8: start-to-start cycles (segment switches and hand-overs stand out)
                live = [i for i in range(64) if int(d[i, 0]) != 0]
                print("      step lengths:", " ".join(str(int(d[live[k + 1], 0] - d[live[k], 0])) for k in range(len(live) - 1)))
print("ALL BIT-IDENTICAL" if bad == 0 else f"{bad} SHAPES DIFFER")
sys.exit(0 if bad == 0 else 1)
