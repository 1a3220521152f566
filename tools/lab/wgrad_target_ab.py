"""Lab: tr_linear_bwd_params (wgrad + reduce) for several token-split targets (libs built with -DTR_WGRAD_TARGET=...), same process."""
import ctypes as C
import glob
import sys
import torch
sys.path.insert(0, ".")
from tokenreduction_amd import _lib

libs = {"1024 (product)": _lib.load()}
for f in sorted(glob.glob("tools/lab/libtr_wg*.so")):
    l = C.CDLL(f)
    l.tr_linear_bwd_params.restype = C.c_int
    l.tr_linear_bwd_params.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                       C.c_int, C.c_int, C.c_int, C.c_void_p]
    l.tr_wgrad_workspace_floats.restype = C.c_size_t
    l.tr_wgrad_workspace_floats.argtypes = [C.c_int] * 3
    libs[f.split("wg")[1].split(".")[0]] = l
tot = {k: 0.0 for k in libs}
for label, M, shapes in (("DeiT-S B=256 N=197", 256 * 197, ((1152, 384), (384, 384), (1536, 384), (384, 1536))),
                         ("DeiT-S B=256 N=97", 256 * 97, ((1152, 384), (384, 384), (1536, 384), (384, 1536))),
                         ("DeiT-B B=128 N=197", 128 * 197, ((2304, 768), (768, 768), (3072, 768), (768, 3072)))):
    for N, K in shapes:
        dy = torch.randn(M, N, device="cuda").bfloat16()
        x = torch.randn(M, K, device="cuda").bfloat16()
        dw = torch.empty(N, K, device="cuda")
        db = torch.empty(N, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        res = {}
        for rep in range(2):
            for key, lib in libs.items():
                nws = max(lib.tr_wgrad_workspace_floats(M, N, K), 1)
                ws = torch.empty(nws, device="cuda")
                args = (dy.data_ptr(), N, 0, x.data_ptr(), K, dw.data_ptr(), db.data_ptr(), 0, ws.data_ptr(), nws, M, N, K, st)
                for _ in range(2):
                    lib.tr_linear_bwd_params(*args)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    lib.tr_linear_bwd_params(*args)
                e1.record()
                torch.cuda.synchronize()
                res[key] = min(res.get(key, 1e9), e0.elapsed_time(e1) * 100)
        for k in tot:
            tot[k] += res[k]
        print(f"{label} dW[{N:4d},{K:4d}]: " + "  ".join(f"{k}: {v:6.1f}" for k, v in res.items()))
print("sum (us):", {k: round(v) for k, v in tot.items()})
