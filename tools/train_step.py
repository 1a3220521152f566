#!/usr/bin/env python3
"""One fine-tune workload for profiling: `python tools/train_step.py [model] [batch] [steps] [torch|hip]` runs fwd + loss + bwd + AdamW
steps of the HIP training path (put it after `rocprofv3 --kernel-trace --stats --`); the last argument picks torch.optim.AdamW(fused=True)
or tokenreduction_amd.optim.FusedAdamW (default)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else bench.MODEL
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
kr = [0.5] if "base" in name else [0.7]
loc = [] if name.startswith("deit") else [3, 6, 9]
torch.cuda.set_device(0)
model = bench.build_model(name, [1.0] if not loc else kr, loc, "cuda").train()
x = torch.randn(batch, 3, 224, 224, device="cuda")
y = torch.randint(0, 1000, (batch,), device="cuda")
which = sys.argv[4] if len(sys.argv) > 4 else "hip"
if which == "hip":
    from tokenreduction_amd.optim import FusedAdamW
    opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.05, model=model)
else:
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05, fused=True)
for i in range(steps + 2):
    if i == 2:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    out = model(x)
    loss = torch.nn.functional.cross_entropy(out[0] if isinstance(out, tuple) else out, y)      # DyViT returns its train tuple (dyvit.py:257-261)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{name} B={batch} AdamW[{which}]: {1e3 * dt:.2f} ms/step, {batch / dt:.0f} images/s, loss {loss.item():.4f}")
