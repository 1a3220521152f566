#!/bin/bash
# Lab: the headline forward (and Top-K keep_rate 0.5) with the eval Mlp as two GEMM launches (fused 0), as the fused launch on the auto policy
# followed by the LayerNorm launch (fused -1, tail 0: the library default), and with the fused block tail (Mlp + residual + next norm1 in one
# launch: tail 1, OFF by default -- it lost 4 % in the model); twice round-robin on one box
python - <<'PY'
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import torch, bench
from tokenreduction_amd import ops
x = torch.randn(bench.BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
for rep in range(2):
    for mode, tail in ((0, 0), (-1, 0), (-1, 1), (0, 0)):
        ops.set_mlp_fused(mode)
        ops.set_mlp_resid_ln(bool(tail))
        for name, kr in (("topk kr0.7", [0.7]), ("topk kr0.5", [0.5])):
            m = bench.build_model(keep_rate=kr)
            ips = bench.quick_images_per_s(m, x, iters=20, reps=3)
            print(f"fused {mode:2d} tail {tail}  {name}: {ips:9.1f} images/s  {bench.BATCH / ips * 1e3:.3f} ms", flush=True)
            del m
PY
