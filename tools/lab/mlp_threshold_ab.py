import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
from tokenreduction_amd import ops
x = torch.randn(bench.BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
for rep in range(3):
    for mode in (257, 300, 0):
        ops.set_mlp_fused(mode)
        m = bench.build_model(keep_rate=[0.7])
        ips = bench.quick_images_per_s(m, x, iters=20, reps=3)
        print(f"mode {mode:3d}: {bench.BATCH / ips * 1e3:.3f} ms", flush=True)
        del m
