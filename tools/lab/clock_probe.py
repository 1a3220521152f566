"""Lab: the shader clock gemm_bf16_pc and mlp_fused_kernel actually run at -- d(s_memtime) / d(s_memrealtime) x 100 MHz around workgroup 8's
main loop, summed over launches (MI355X_MICROARCH.md, DVFS item 6) -- (a) inside the headline forward under hipGraph replay, after
>= 2.5 s of back-to-back replays, (b) each kernel alone, back to back, at the headline's first-stage shape.
Needs the -DTR_DIAG_CLOCK build:  tools/lab/build_clock.sh;  TOKENREDUCTION_HIP_LIB=$PWD/tools/lab/libtr_clock.so python tools/lab/clock_probe.py [out.json]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from tokenreduction_amd import _lib, ops

lib = _lib.load()


def read(which):
    buf = (C.c_ulonglong * 3)()
    _lib.check((lib.tr_gemm_clock_probe_read if which == "gemm" else lib.tr_mlp_clock_probe_read)(buf), "clock_probe_read")
    cyc, ticks, n = int(buf[0]), int(buf[1]), int(buf[2])
    return dict(launches=n, shader_cycles=cyc, us=round(ticks / 100.0, 1), mhz=round(cyc / (ticks / 100.0)) if ticks else None)


def burn(fn, seconds):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    return n


rec = {"method": "d(s_memtime)/d(s_memrealtime) x 100 MHz of workgroup 8, summed over the launches between two reads; -DTR_DIAG_CLOCK build",
       "device": torch.cuda.get_device_name(0)}
x = torch.randn(bench.BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
model = bench.build_model()
burn(lambda: model(x), 2.5)
read("gemm"), read("mlp")
samples = []
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(100):
        model(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 10.0
    samples.append({"ms_per_forward": round(ms, 3), "gemm_bf16_pc": read("gemm"), "mlp_fused_kernel": read("mlp")})
rec["headline_forward_hipgraph_replay"] = samples
if read("gemm")["launches"] == 0 and samples[0]["gemm_bf16_pc"]["launches"] == 0:
    print("this library was not built with -DTR_DIAG_CLOCK (tools/lab/build_clock.sh)", file=sys.stderr)

g = torch.Generator().manual_seed(1)
M, D, Hd = 50432, 384, 1536
xn = torch.randn(M, D, generator=g).bfloat16().cuda()
wq, bq = (0.05 * torch.randn(3 * D, D, generator=g)).bfloat16().cuda(), torch.zeros(3 * D).cuda()
w1, w2 = (0.05 * torch.randn(Hd, D, generator=g)).bfloat16().cuda(), (0.05 * torch.randn(D, Hd, generator=g)).bfloat16().cuda()
b1, b2 = torch.zeros(Hd).cuda(), torch.zeros(D).cuda()
pk = ops.mlp_pack(w1, w2, b2)
oq, om = torch.empty(M, 3 * D, dtype=torch.bfloat16, device="cuda"), torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
alone = {}
for name, which, fn in (("gemm_bf16_pc qkv 50432x1152x384", "gemm", lambda: ops.gemm(xn, wq, bq, ops.TR_EPI_BF16, out=oq)),
                        ("mlp_fused_kernel 50432 rows", "mlp", lambda: ops.mlp_fused(xn, pk, b1, out=om))):
    burn(fn, 2.5)
    read(which)
    burn(fn, 0.5)
    alone[name] = read(which)
rec["alone_back_to_back"] = alone
out = json.dumps(rec, indent=1)
print(out)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(out + "\n")
