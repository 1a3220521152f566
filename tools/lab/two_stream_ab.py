#!/usr/bin/env python3
"""Lab: the headline forward as ONE batch of 256 vs TWO half batches of 128 on two streams (two model instances with the same weights, each
with its own workspace / captured graph) -- do the tails of one stream's kernels get filled by the other stream's?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda")
x = torch.randn(bench.BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(0)).to(dev)
m = bench.build_model(keep_rate=[0.7])
ips = bench.quick_images_per_s(m, x, iters=20, reps=3)
print(f"one batch of {bench.BATCH}: {bench.BATCH / ips * 1e3:.3f} ms", flush=True)
ma, mb = bench.build_model(keep_rate=[0.7]), bench.build_model(keep_rate=[0.7])
mb.load_state_dict(ma.state_dict())
xa, xb = x[:128].contiguous(), x[128:].contiguous()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for mode in ("two streams", "one stream"):
    def step():
        if mode == "two streams":
            with torch.cuda.stream(sa):
                ma(xa)
            with torch.cuda.stream(sb):
                mb(xb)
        else:
            ma(xa); mb(xb)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    print(f"two half batches, {mode}: {best * 1e3:.3f} ms per 256 images", flush=True)
