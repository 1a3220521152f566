#!/usr/bin/env python3
"""Run a few forwards of one registered model (for rocprofv3):  python3 tools/run_model.py tome_small_patch16_224 r16 [batch] [iters] [img_size] [precision]
keep spec: 'r16' (ToMe: 16 merged per block, every block) or a float keep_rate with reduction_loc 3,6,9; precision bf16 (default) | bf16x3 | fp32."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import build_model  # noqa: E402

name, spec = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
img = int(sys.argv[5]) if len(sys.argv) > 5 else 224
if spec.startswith("r"):
    r = int(spec[1:])
    kr, loc = [196 - r * (i + 1) for i in range(12)], list(range(12))
else:
    kr, loc = [float(spec)], [3, 6, 9]
m = build_model(name, kr, loc, "cuda", img_size=img)
if len(sys.argv) > 6:
    m.precision = sys.argv[6]
x = torch.randn(B, 3, img, img, generator=torch.Generator().manual_seed(1)).cuda()
for _ in range(3):
    m(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    m(x)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{name} {spec} B={B}: {B * iters / dt:.1f} images/s, {1e3 * dt / iters:.3f} ms/forward, tokens {m._last_tokens}")
