"""Lab: A/B of tr_attention_bwd_bf16 between two builds, interleaved in one process.  usage: python tools/lab/attn_bwd_ab.py old.so"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from tokenreduction_amd import _lib

old = C.CDLL(sys.argv[1])
new = _lib.load()
old.tr_attention_bwd_bf16.restype = C.c_int
old.tr_attention_bwd_bf16.argtypes = [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p]
B, H = 256, 6
for N in (197, 138, 97, 68):
    qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 1.5).bfloat16()
    do = torch.randn(B * N, H * 64, device="cuda").bfloat16()
    outs = {k: torch.empty_like(qkv) for k in ("old", "new")}
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for rep in range(3):
        for key, lib in (("old", old), ("new", new)):
            for _ in range(2):
                lib.tr_attention_bwd_bf16(qkv.data_ptr(), do.data_ptr(), None, None, outs[key].data_ptr(), B, N, H, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.tr_attention_bwd_bf16(qkv.data_ptr(), do.data_ptr(), None, None, outs[key].data_ptr(), B, N, H, st)
            e1.record()
            torch.cuda.synchronize()
            res[key] = min(res.get(key, 1e9), e0.elapsed_time(e1) * 1e3 / 20)
    print(f"N={N:3d}: old {res['old']:7.1f} us  new {res['new']:7.1f} us ({100 * (res['new'] / res['old'] - 1):+5.1f} %)  identical {torch.equal(outs['old'], outs['new'])}")
