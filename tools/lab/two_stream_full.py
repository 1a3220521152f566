#!/usr/bin/env python3
"""Lab: TWO full batches of 256 in flight on two streams (two model instances with the same weights, each with its own workspace and captured
graph) against the same forwards one after the other on one stream -- do the tails of one forward's launches (partial last rounds of the
persistent kernels, the small kernels of the last stage) get filled by the other's?  (tools/lab/two_stream_ab.py halved the batch: slower.)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda")
xa = torch.randn(bench.BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(0)).to(dev)
xb = torch.randn(bench.BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(dev)
for kr in ([0.7], [0.5]):
    ma, mb = bench.build_model(keep_rate=kr), bench.build_model(keep_rate=kr)
    mb.load_state_dict(ma.state_dict())
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    res = {}
    for mode in ("one stream", "two streams", "one stream", "two streams"):
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
            best = min(best, (time.perf_counter() - t0) / 40)
        print(f"keep_rate {kr[0]}: two batches of 256, {mode}: {best * 1e3:.3f} ms per batch = {256 / best:.0f} images/s", flush=True)
    del ma, mb
