#!/usr/bin/env python3
"""Forward latency at small batch (BASELINE configs[0]: topk_small kr0.9 B=8), eager launches vs hipGraph replay."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import build_model  # noqa: E402

for B in (1, 8, 32):
    m = build_model("topk_small_patch16_224", [0.9], [3, 6, 9], "cuda")
    x = torch.randn(B, 3, 224, 224).cuda()
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        m(x)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 100
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        m(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = m(x)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 100
    print(f"B={B:3d}: eager {1e3 * eager:.3f} ms ({B / eager:.0f} img/s)   hipGraph replay {1e3 * graph:.3f} ms ({B / graph:.0f} img/s)")
