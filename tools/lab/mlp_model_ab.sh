#!/bin/bash
# Lab: the headline forward with the eval Mlp as two GEMM launches (mode 0), and on the auto policy with the stream-K launch taken from 257 /
# 300 / 400 blocks of 128 rows on (tr_set_mlp_fused(mode >= 2); the product default is 257: every launch of more than one round)
python - <<'PY'
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import torch, bench
from tokenreduction_amd import ops
x = torch.randn(bench.BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
for rep in range(2):
    for mode in (0, 257, 300, 400, 0):
        ops.set_mlp_fused(mode)
        for name, kr in (("topk kr0.7", [0.7]), ("topk kr0.5", [0.5])):
            m = bench.build_model(keep_rate=kr)
            ips = bench.quick_images_per_s(m, x, iters=20, reps=3)
            print(f"mode {mode:2d}  {name}: {ips:9.1f} images/s  {bench.BATCH / ips * 1e3:.3f} ms", flush=True)
            del m
PY
