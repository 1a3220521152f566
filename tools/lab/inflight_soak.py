"""Lab: soak of model.forward_async -- many forwards, two in flight, alternating inputs, uneven load on the caller's stream and a third stream;
every result compared bit for bit with model(x).  usage: python tools/lab/inflight_soak.py [launches]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
random.seed(0)
bad = 0
for name, kr in (("topk_small_patch16_224", [0.7]), ("tome_small_patch16_224", [196 - 16 * (i + 1) for i in range(12)]), ("evit_small_patch16_224", [0.5])):
    m = bench.build_model(name, kr, list(range(12)) if "tome" in name else [3, 6, 9])
    xs = [torch.randn(bench.BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(s)).cuda() for s in (1, 2, 3)]
    want = [m(x).clone() for x in xs]
    side, junk = torch.cuda.Stream(), torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    pend = []
    for k in range(n):
        i = random.randrange(3)
        pend.append((m.forward_async(xs[i]), i))
        if random.random() < 0.3:
            with torch.cuda.stream(side):
                junk.add_(1)
        if random.random() < 0.2:
            junk[: 1 << 20].add_(1)
        while len(pend) > random.choice((1, 2, 3)):
            h, j = pend.pop(0)
            bad += 0 if torch.equal(h.result(), want[j]) else 1
    for h, j in pend:
        bad += 0 if torch.equal(h.result(), want[j]) else 1
    torch.cuda.synchronize()
    m.check_status()
    print(f"{name}: {n} forwards through forward_async, {bad} differ from model(x)", flush=True)
    del m
sys.exit(1 if bad else 0)
