#!/usr/bin/env python3
"""Dev experiment: one batch as two half batches on two HIP streams (two executors, own workspaces) vs one full-batch call.
Every kernel of a forward depends on the previous one, so the only overlap available is between the two halves: one half's
HBM-bound LayerNorm / tail round under the other half's GEMM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model

name = sys.argv[1] if len(sys.argv) > 1 else "topk_small_patch16_224"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
kr, loc = [0.7], [3, 6, 9]
full = build_model(name, kr, loc, "cuda")
halves = [build_model(name, kr, loc, "cuda") for _ in range(2)]
x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(1)).cuda()
xs = [x[:B // 2].contiguous(), x[B // 2:].contiguous()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run_full():
    return full(x)


def run_split():
    outs = []
    cur = torch.cuda.current_stream()
    for m, xi, s in zip(halves, xs, streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(m(xi))
    for s in streams:
        cur.wait_stream(s)
    return outs


for fn, label in ((run_full, "one call, B=%d" % B), (run_split, "two streams, 2 x %d" % (B // 2)), (run_full, "one call again")):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    print(f"{label:28s}: {1e3 * best:.3f} ms per batch, {B / best:.0f} images/s")
a = run_full()
b = torch.cat(run_split())
torch.cuda.synchronize()
print("max |full - split| logits:", (a.float() - b.float()).abs().max().item())
