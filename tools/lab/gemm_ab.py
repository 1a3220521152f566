"""Lab: A/B of two builds of the library on the GEMM shapes of the headline forward, interleaved in one process (boxes differ by
up to 10 %, so only same-run comparisons count).  usage: python tools/lab/gemm_ab.py tools/lab/libtr_old.so"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from tokenreduction_amd import _lib, ops

old = C.CDLL(sys.argv[1])
new = _lib.load()
for l in (old,):
    l.tr_gemm_bf16.restype = C.c_int
    l.tr_gemm_bf16.argtypes = [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p]
dev = "cuda"
tot = {"old": 0.0, "new": 0.0}
for tokens in (197, 138, 97, 68):
    M = 256 * tokens
    for name, N, K, epi in (("qkv", 1152, 384, 0), ("proj", 384, 384, 0), ("fc1", 1536, 384, 1), ("fc2", 384, 1536, 0)):
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.zeros(N, device=dev)
        outs = {k: torch.empty(M, N, device=dev, dtype=torch.bfloat16) for k in ("old", "new")}
        st = torch.cuda.current_stream().cuda_stream
        res = {}
        for rep in range(3):
            for key, lib in (("old", old), ("new", new)):
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
        same = torch.equal(outs["old"], outs["new"])
        for k in tot:
            tot[k] += 3 * res[k]
        print(f"tokens {tokens:3d} {name:4s}: old {res['old']:7.2f} us  new {res['new']:7.2f} us  ({100 * (res['new'] / res['old'] - 1):+5.1f} %)  bit-identical: {same}")
print(f"GEMM time per forward (3 blocks per stage): old {tot['old']:.0f} us, new {tot['new']:.0f} us")
