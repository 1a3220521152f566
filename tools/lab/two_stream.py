"""Lab: does running the batch as independent sub-batches on separate HIP streams fill the GEMM tails / overlap the HBM-bound
LayerNorms with the MFMA-bound GEMMs?  DeiT-S Top-K kr 0.7, B = 256 total."""
import sys
import time
import torch
sys.path.insert(0, ".")
import bench

dev = "cuda"
x = torch.randn(256, 3, 224, 224, generator=torch.Generator().manual_seed(7)).to(dev)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


m = bench.build_model()
m.use_graph = False
print(f"one stream, B=256, plain launches: {timed(lambda: m(x)):.3f} ms")
m.use_graph = True
print(f"one stream, B=256, graph replay:   {timed(lambda: m(x)):.3f} ms")

for parts in (2, 4):
    models = [bench.build_model() for _ in range(parts)]
    for mm in models:
        mm.use_graph = False
    streams = [torch.cuda.Stream() for _ in range(parts)]
    xs = [c.contiguous() for c in x.chunk(parts)]

    def run():
        cur = torch.cuda.current_stream()
        for mm, s, xx in zip(models, streams, xs):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                mm(xx)
        for s in streams:
            cur.wait_stream(s)
    print(f"{parts} streams x B={256 // parts}, plain launches: {timed(run):.3f} ms")
    # sequential sub-batches on ONE stream (what the split alone costs)
    def run_seq():
        for mm, xx in zip(models, xs):
            mm(xx)
    print(f"{parts} sub-batches of {256 // parts} on one stream:      {timed(run_seq):.3f} ms")
