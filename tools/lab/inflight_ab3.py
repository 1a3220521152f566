"""Lab: two forwards in flight -- (a) norm2 inside the fused Mlp (tr_set_mlp_ln 0 / 1 / 2) now that `concurrent` launches run whole blocks,
(b) run with TOKENREDUCTION_HIP_LIB=tools/lab/libtr_gemm_v152.so: gemm_bf16_pc on 152 registers, which leaves room for one LayerNorm wave
per SIMD beside it (see coresidency_probe.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from tokenreduction_amd import ops
x = torch.randn(bench.BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
tag = os.path.basename(os.environ.get("TOKENREDUCTION_HIP_LIB", "product"))
modes = [int(v) for v in sys.argv[1:]] or [1]
for rep in range(2):
    for mode in modes:
        ops.set_mlp_ln(mode)
        for name, kr in (("kr0.7", [0.7]), ("kr0.5", [0.5]), ("dense", None)):
            m = bench.build_model(keep_rate=kr) if kr else bench.build_model("deit_small_patch16_224_local", [1.0], [])
            one = bench.quick_images_per_s(m, x, iters=20, reps=3, in_flight=1)
            two = bench.quick_images_per_s(m, x, iters=20, reps=3, in_flight=2)
            print(f"{tag} ln-mode {mode} {name}: one at a time {one:9.0f}   two in flight {two:9.0f} images/s", flush=True)
            del m
ops.set_mlp_ln(1)
