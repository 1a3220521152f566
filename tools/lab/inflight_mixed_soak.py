"""Lab: model.forward_async with batches of DIFFERENT sizes and inputs at changing addresses in flight (one workspace + graph per (batch size,
side stream); an input at a new address re-captures or falls back to plain launches), interleaved with model(x) on the caller's stream;
every result compared bit for bit with a reference taken up front."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
random.seed(1)
total_bad = 0
for name, kr, loc in (("topk_small_patch16_224", [0.7], [3, 6, 9]), ("dpcknn_small_patch16_224", [0.7], [3, 6, 9]), ("ats_small_patch16_224", [0.7], [3, 6, 9]),
                      ("tome_small_patch16_224", [196 - 16 * (i + 1) for i in range(12)], list(range(12)))):
    m = bench.build_model(name, kr, loc)
    sizes = (256, 64, 7, 130)
    xs = {B: torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(B)).cuda() for B in sizes}
    if hasattr(m, "density_noise"):
        shapes = m._stage_shapes()
    want = {}
    for B in sizes:
        if hasattr(m, "density_noise"):
            m.density_noise = {blk: torch.rand(B, P, generator=torch.Generator().manual_seed(blk)) for blk, _, P in shapes}
        want[B] = m(xs[B]).clone()
    bad, pend = 0, []
    for k in range(n):
        B = random.choice(sizes)
        if hasattr(m, "density_noise"):
            m.density_noise = {blk: torch.rand(B, P, generator=torch.Generator().manual_seed(blk)) for blk, _, P in shapes}
        x = xs[B] if random.random() < 0.7 else xs[B].clone()            # sometimes at a fresh address
        if random.random() < 0.25:
            bad += 0 if torch.equal(m(x), want[B]) else 1                 # a plain forward on the caller's stream in between
        else:
            pend.append((m.forward_async(x), B))
        while len(pend) > random.choice((0, 1, 2)):
            h, b = pend.pop(0)
            bad += 0 if torch.equal(h.result(), want[b]) else 1
    for h, b in pend:
        bad += 0 if torch.equal(h.result(), want[b]) else 1
    torch.cuda.synchronize()
    m.check_status()
    total_bad += bad
    print(f"{name}: {n} forwards of batch sizes {sizes} mixed, {bad} differ", flush=True)
    del m
print("ALL OK" if total_bad == 0 else f"{total_bad} DIFFER")
sys.exit(1 if total_bad else 0)
