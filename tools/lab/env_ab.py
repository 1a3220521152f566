"""Lab: headline forward (and Top-K kr 0.5, dense DeiT-S) with an environment switch off / on, one subprocess per arm, round-robin.
usage: python tools/lab/env_ab.py VAR=value [VAR2=value ...]   (each given assignment is one arm; the first arm is always 'no switch')"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, time, torch
sys.path.insert(0, %r)
import bench
x = torch.randn(256, 3, 224, 224, device="cuda")
out = []
for name, kr in (("kr0.7", [0.7]), ("kr0.5", [0.5]), ("dense", None)):
    m = bench.build_model(keep_rate=kr) if kr else bench.build_model("deit_small_patch16_224_local", [1.0], [])
    for _ in range(5): m(x)
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): m(x)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 20)
    out.append("%%s %%.4f ms" %% (name, best * 1e3))
    del m
print("   ".join(out))
''' % ROOT
arms = [("(default)", {})] + [(a, dict([a.split("=", 1)])) for a in sys.argv[1:]]
for rnd in range(3):
    for name, env in arms:
        o = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
        print(f"{name:28s}", o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-300:], flush=True)
