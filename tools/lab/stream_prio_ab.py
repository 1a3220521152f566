import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
x = torch.randn(bench.BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
dev = x.device
for name, prios in (("default/default", (0, 0)), ("high/low", (-1, 0)), ("high/high", (-1, -1)), ("default/default", (0, 0)), ("high/low", (-1, 0))):
    m = bench.build_model()
    m.__dict__["_pipe_streams"] = {dev: [torch.cuda.Stream(device=dev, priority=p) for p in prios]}
    two = bench.quick_images_per_s(m, x, iters=20, reps=3, in_flight=2)
    print(f"{name}: {two:9.0f} images/s", flush=True)
    del m
