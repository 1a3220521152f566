"""Lab: race screen of the EVAL path over every family at full width: model(x) n times back to back (graph replay, no host synchronisation in
between) and n times through model.forward_async with two in flight; every logit tensor must equal the first forward's bit for bit (the eval
kernels sum in a fixed order; DPC-KNN's density noise is drawn once and replayed).  tools/lab/eval_soak.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
PREC = sys.argv[2] if len(sys.argv) > 2 else "bf16"          # bf16 | bf16x3 | fp32 (the tolerance-conformant and the fp32 executors: smaller batches)
CASES = [(f"{f}_small_patch16_224", [0.7], [3, 6, 9], 224, 256) for f in ("topk", "evit", "dyvit", "sit", "ats", "dpcknn", "sinkhorn", "kmedoids", "patchmerger", "heuristic")]
CASES += [("tome_small_patch16_224", [196 - 16 * (i + 1) for i in range(12)], list(range(12)), 224, 256), ("deit_small_patch16_224_local", [1.0], [], 224, 256),
          ("topk_small_patch16_224", [0.5], [3, 6, 9], 224, 256), ("ats_base_patch16_224", [0.5], [3, 6, 9], 224, 128),
          ("dpcknn_base_patch16_224", [0.5], [3, 6, 9], 224, 128), ("sinkhorn_base_patch16_224", [0.25], [3, 6, 9], 384, 64),
          ("kmedoids_base_patch16_224", [0.25], [3, 6, 9], 384, 64), ("topk_base_patch16_224", [0.7], [3, 6, 9], 224, 128)]
total_bad = 0
for name, kr, loc, img, B in CASES:
    m = bench.build_model(name, kr, loc, "cuda", img)
    if PREC != "bf16":
        m.precision = PREC
        B = max(16, B // 4)
    x = torch.randn(B, 3, img, img, generator=torch.Generator().manual_seed(3)).cuda()
    if hasattr(m, "density_noise"):                      # DPC-KNN draws fresh density noise every forward (dpcknn.py:71-72): pinned for the comparison
        m.density_noise = {blk: torch.rand(B, P, generator=torch.Generator().manual_seed(blk)) for blk, _, P in m._stage_shapes()}
    with torch.no_grad():
        first = m(x)
        first = (first[0] if isinstance(first, tuple) else first).clone()
        bad_seq = bad_async = 0
        outs = []
        for _ in range(n):
            o = m(x)
            outs.append((o[0] if isinstance(o, tuple) else o).clone())
        bad_seq = sum(0 if torch.equal(o, first) else 1 for o in outs)
        if not getattr(m, "dynamic_width", False):
            pend = []
            for _ in range(n):
                pend.append(m.forward_async(x))
                if len(pend) > 1:
                    o = pend.pop(0).result()
                    bad_async += 0 if torch.equal(o[0] if isinstance(o, tuple) else o, first) else 1
            for h in pend:
                o = h.result()
                bad_async += 0 if torch.equal(o[0] if isinstance(o, tuple) else o, first) else 1
        torch.cuda.synchronize()
        m.check_status()
    total_bad += bad_seq + bad_async
    print(f"{PREC} {name} kr={kr[0]} {img}^2 B={B}: {n} forwards one at a time: {bad_seq} differ; {n} with two in flight: {bad_async} differ", flush=True)
    del m, outs
print("ALL OK" if total_bad == 0 else f"{total_bad} DIFFER")
sys.exit(1 if total_bad else 0)
