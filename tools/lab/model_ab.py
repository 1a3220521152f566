"""Lab: headline forward (bench.py model, batch 256) through several builds of the library, interleaved in one process.
usage: python tools/lab/model_ab.py name=path.so [...]   (TOKENREDUCTION_HIP_LIB is read by tokenreduction_amd._lib at load time: one subprocess per build)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, time, torch
sys.path.insert(0, %r)
import bench
m = bench.build_model()
x = torch.randn(256, 3, 224, 224, device="cuda")
for _ in range(5): m(x)
best = 1e9
for rep in range(5):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): m(x)
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 20)
print("%%.4f ms" %% (best * 1e3))
''' % ROOT
for rnd in range(2):
    for a in sys.argv[1:]:
        n, p = a.split("=")
        env = dict(os.environ, TOKENREDUCTION_HIP_LIB=os.path.abspath(p))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(n, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
