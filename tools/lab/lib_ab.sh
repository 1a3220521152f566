#!/bin/bash
# Lab: the same workloads on the product library and on a variant build (TOKENREDUCTION_HIP_LIB), alternating processes on one box
#   tools/lab/lib_ab.sh tools/lab/libtr_<name>.so
V=$1
cat > /tmp/lib_ab.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
x = torch.randn(bench.BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
m = bench.build_model(keep_rate=[0.7])
ips = bench.quick_images_per_s(m, x, iters=20, reps=3)
print(f"  headline forward {bench.BATCH / ips * 1e3:.3f} ms", flush=True)
PY
for rep in 1 2 3; do
  echo "product:"; python /tmp/lib_ab.py 2>&1 | grep -v amdgpu.ids
  echo "variant $V:"; TOKENREDUCTION_HIP_LIB=$V python /tmp/lib_ab.py 2>&1 | grep -v amdgpu.ids
done
for rep in 1 2; do
  echo "product train:"; python tools/train_step.py topk_small_patch16_224 256 10 2>&1 | tail -1
  echo "variant train:"; TOKENREDUCTION_HIP_LIB=$V python tools/train_step.py topk_small_patch16_224 256 10 2>&1 | tail -1
done
