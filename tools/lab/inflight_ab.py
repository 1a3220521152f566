"""Lab: with two forwards in flight (model.forward_async) the other forward fills a launch's tail -- do the choices that were tuned one forward
at a time still hold?  Headline model, per setting: images/s one at a time and two in flight."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from tokenreduction_amd import ops
x = torch.randn(bench.BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
settings = [("product (fused Mlp auto, norm2 inside one-round launches)", -1, 1), ("Mlp as two GEMMs everywhere", 0, 1), ("fused Mlp everywhere", 1, 1),
            ("norm2 never inside", -1, 0), ("norm2 inside wherever fused", -1, 2), ("stream-K only from 300 blocks", 300, 1)]
for rep in range(2):
    for name, fused, ln in settings:
        ops.set_mlp_fused(fused)
        ops.set_mlp_ln(ln)
        m = bench.build_model()
        one = bench.quick_images_per_s(m, x, iters=20, reps=3, in_flight=1)
        two = bench.quick_images_per_s(m, x, iters=20, reps=3, in_flight=2)
        print(f"{name:58s} one at a time {one:9.0f}   two in flight {two:9.0f} images/s", flush=True)
        del m
ops.set_mlp_fused(-1); ops.set_mlp_ln(1)
