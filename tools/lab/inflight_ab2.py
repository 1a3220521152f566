"""Lab: two forwards in flight -- the executor's `concurrent` hint (fused Mlp wherever supported) and the stream-K schedule under it.
Run once per environment (TR_MLP_NO_STREAMK=1: whole blocks round-robin, no hand-over)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
x = torch.randn(bench.BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
for name, kr in (("kr0.7", [0.7]), ("kr0.5", [0.5]), ("dense", None)):
    m = bench.build_model(keep_rate=kr) if kr else bench.build_model("deit_small_patch16_224_local", [1.0], [])
    one = bench.quick_images_per_s(m, x, iters=20, reps=3, in_flight=1)
    two = bench.quick_images_per_s(m, x, iters=20, reps=3, in_flight=2)
    print(f"{os.environ.get('TR_MLP_NO_STREAMK', '0')} {name}: one at a time {one:9.0f}   two in flight {two:9.0f} images/s", flush=True)
