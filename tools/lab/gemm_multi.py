"""Lab: same-process comparison of several builds of the library on the GEMM shapes of the headline forward.
usage: python tools/lab/gemm_multi.py name=path.so [name=path.so ...]   (the first one is the reference for bit-identity)"""
import ctypes as C
import sys
import torch

libs = []
for a in sys.argv[1:]:
    n, p = a.split("=")
    l = C.CDLL(p)
    l.tr_gemm_bf16.restype = C.c_int
    l.tr_gemm_bf16.argtypes = [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p]
    libs.append((n, l))
dev = "cuda"
tot = {n: 0.0 for n, _ in libs}
shapes = (("qkv", 1152, 384, 0), ("proj", 384, 384, 0), ("fc1", 1536, 384, 1), ("fc2", 384, 1536, 0))
for tokens in (197, 138, 97, 68):
    M = 256 * tokens
    for name, N, K, epi in shapes:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(N, device=dev)
        outs = {n: torch.zeros(M, N, device=dev, dtype=torch.bfloat16) for n, _ in libs}
        st = torch.cuda.current_stream().cuda_stream
        res = {}
        for rep in range(3):
            for key, lib in libs:
                for _ in range(3):
                    lib.tr_gemm_bf16(a.data_ptr(), w.data_ptr(), b.data_ptr(), outs[key].data_ptr(), None, 0, M, N, K, epi, st)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(40):
                    lib.tr_gemm_bf16(a.data_ptr(), w.data_ptr(), b.data_ptr(), outs[key].data_ptr(), None, 0, M, N, K, epi, st)
                e1.record()
                torch.cuda.synchronize()
                res[key] = min(res.get(key, 1e9), e0.elapsed_time(e1) * 1e3 / 40)
        ref = libs[0][0]
        line = f"tokens {tokens:3d} {name:4s}:"
        for key, _ in libs:
            tot[key] += 3 * res[key]
            same = "=" if torch.equal(outs[ref], outs[key]) else "x"
            line += f"  {key} {res[key]:7.2f}{same}"
        print(line, f" [{2.0 * M * N * K / res[ref] / 1e6:.0f} TF ref]")
print("GEMM time per forward (3 blocks per stage):", {k: round(v) for k, v in tot.items()})
