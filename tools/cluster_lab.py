#!/usr/bin/env python3
"""Dev tool: the DPC-KNN clustering of one stage, one launch (fast_dist=1) against the staged launches (fast_dist=2), at the stage shapes
of dpcknn_small / dpcknn_base at batch 256 / 128:  python3 tools/cluster_lab.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tokenreduction_amd import ops  # noqa: E402

for B, N, D, K in ((256, 197, 384, 137), (256, 138, 384, 96), (256, 97, 384, 67), (128, 197, 768, 98), (128, 99, 768, 49), (128, 50, 768, 24)):
    x = torch.randn(B, N, D, device="cuda")
    noise = torch.rand(B, N - 1, device="cuda")
    res = {}
    for mode in (2, 1):
        for _ in range(3):
            out = ops.dpcknn_cluster(x, K, noise, 5, fast_dist=mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = ops.dpcknn_cluster(x, K, noise, 5, fast_dist=mode)
        torch.cuda.synchronize()
        res[mode] = (1e6 * (time.perf_counter() - t0) / 20, out)
    if os.environ.get("FT_STAMPS"):
        import ctypes
        from tokenreduction_amd import _lib
        buf = (ctypes.c_ulonglong * 8)()
        _lib.load().ftdbg_read(buf)
        t = list(buf)
        print("   cycles: norms %d, gram %d, dist->LDS %d, density %d, dmax+parent %d, top-K %d, assign+out %d | total %d" %
              (t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6], t[7] - t[0]))
    same_c = torch.equal(res[1][1][0], res[2][1][0])
    same_a = float((res[1][1][1] == res[2][1][1]).float().mean())
    print(f"B={B} N={N} D={D} K={K}: staged {res[2][0]:7.1f} us   one launch {res[1][0]:7.1f} us   centres equal {same_c}, assignments equal {same_a:.6f}")
