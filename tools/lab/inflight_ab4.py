"""Lab: two forwards in flight -- the fused block tail (tr_set_mlp_resid_ln 1: Mlp + residual + next norm1 in one launch, after a norm2 launch)
against the product policy (norm2 inside the Mlp, norm1 as a launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from tokenreduction_amd import ops
x = torch.randn(bench.BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
for rep in range(2):
    for tail in (0, 1):
        ops.set_mlp_resid_ln(tail)
        for name, kr in (("kr0.7", [0.7]), ("kr0.5", [0.5]), ("dense", None)):
            m = bench.build_model(keep_rate=kr) if kr else bench.build_model("deit_small_patch16_224_local", [1.0], [])
            one = bench.quick_images_per_s(m, x, iters=20, reps=3, in_flight=1)
            two = bench.quick_images_per_s(m, x, iters=20, reps=3, in_flight=2)
            print(f"fused tail {tail} {name}: one at a time {one:9.0f}   two in flight {two:9.0f} images/s", flush=True)
            del m
ops.set_mlp_resid_ln(0)
