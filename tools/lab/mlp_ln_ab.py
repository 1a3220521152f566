"""Lab: norm2 inside the fused Mlp launch (tr_mlp_fused_ln_bf16) against LayerNorm launch + fused Mlp launch.
(1) op level, HIP-event us per call at the model's stage shapes; (2) the headline forward and Top-K kr 0.5, hipGraph replay, round-robin on one box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from tokenreduction_amd import ops

D, Hd = 384, 1536
g = torch.Generator().manual_seed(1)
w1, w2 = (0.05 * torch.randn(Hd, D, generator=g)).bfloat16().cuda(), (0.05 * torch.randn(D, Hd, generator=g)).bfloat16().cuda()
b1, b2 = (0.1 * torch.randn(Hd, generator=g)).cuda(), (0.1 * torch.randn(D, generator=g)).cuda()
ga, be = (1 + 0.2 * torch.randn(D, generator=g)).cuda(), (0.1 * torch.randn(D, generator=g)).cuda()
pk = ops.mlp_pack(w1, w2, b2)


def ev_us(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / n)
    return best


if "--model-only" not in sys.argv:
    for M in (24832, 35328, 50432, 32768, 70001):
        x = (2 * torch.randn(M, D, generator=g)).cuda()
        d = torch.randn(M, D, generator=g).bfloat16().cuda()
        xn = ops.layernorm2(x, ga, be, 1e-6, d, write_x=False)
        out = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
        t_ln = ev_us(lambda: ops.layernorm2(x, ga, be, 1e-6, d, write_x=False))
        t_mlp = ev_us(lambda: ops.mlp_fused(xn, pk, b1, out=out))
        t_two = ev_us(lambda: (ops.layernorm2(x, ga, be, 1e-6, d, write_x=False), ops.mlp_fused(xn, pk, b1, out=out)))
        t_one = ev_us(lambda: ops.mlp_fused_ln(x, d, ga, be, 1e-6, pk, b1, out=out))
        print(f"M={M:6d}: layernorm2 {t_ln:6.1f} us, fused Mlp {t_mlp:6.1f} us, both {t_two:6.1f} us | one launch {t_one:6.1f} us "
              f"({4.0 * M * D * Hd / t_one / 1e6:.0f} TFLOP/s)", flush=True)

x = torch.randn(bench.BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
for rep in range(2):
    for on in (0, 1, 2, 0):
        ops.set_mlp_ln(on)
        for name, kr in (("topk kr0.7", [0.7]), ("topk kr0.5", [0.5]), ("dense", None)):
            m = bench.build_model(keep_rate=kr) if kr else bench.build_model("deit_small_patch16_224_local", [1.0], [])
            ips = bench.quick_images_per_s(m, x, iters=20, reps=3)
            print(f"norm2 in Mlp {int(on)}  {name}: {ips:9.1f} images/s  {bench.BATCH / ips * 1e3:.3f} ms", flush=True)
            del m
ops.set_mlp_ln(1)
