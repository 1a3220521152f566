"""Lab: same-process comparison of several builds of the library on the attention forward.
usage: python tools/lab/attn_multi.py name=path.so [...]"""
import ctypes as C
import sys
import torch

libs = []
for a in sys.argv[1:]:
    n, p = a.split("=")
    l = C.CDLL(p)
    l.tr_attention_bf16.restype = C.c_int
    l.tr_attention_bf16.argtypes = [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p]
    libs.append((n, l))
B, H = 256, 6
for N in (197, 138, 97, 68):
    qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 1.5).bfloat16()
    outs = {n: torch.zeros(B * N, H * 64, device="cuda", dtype=torch.bfloat16) for n, _ in libs}
    cls = {n: torch.zeros(B, H, N, device="cuda") for n, _ in libs}
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for rep in range(3):
        for key, lib in libs:
            for _ in range(3):
                lib.tr_attention_bf16(qkv.data_ptr(), outs[key].data_ptr(), cls[key].data_ptr(), None, None, B, N, H, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                lib.tr_attention_bf16(qkv.data_ptr(), outs[key].data_ptr(), cls[key].data_ptr(), None, None, B, N, H, st)
            e1.record()
            torch.cuda.synchronize()
            res[key] = min(res.get(key, 1e9), e0.elapsed_time(e1) * 1e3 / 40)
    ref = libs[0][0]
    line = f"N {N:3d}:"
    for key, _ in libs:
        d = (outs[key].float() - outs[ref].float()).abs().max().item()
        dc = (cls[key] - cls[ref]).abs().max().item()
        line += f"  {key} {res[key]:6.2f} us (d {d:.1e} cls {dc:.1e})"
    print(line)
