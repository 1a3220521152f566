"""Lab: how far is tr_gemm_bf16 from the vendor library on the forward's shapes?  torch's F.linear (hipBLASLt / rocBLAS, whichever torch
picks) against tr_gemm_bf16, same process, same buffers, bias included, bf16 out.  Comparison only: the product never calls the library."""
import sys
import torch
sys.path.insert(0, ".")
from tokenreduction_amd import ops

dev = "cuda"


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tot_l = tot_o = 0.0
for tokens in (197, 138, 97, 68):
    M = 256 * tokens
    for name, N, K in (("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536)):
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.zeros(N, device=dev)
        bb = b.bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        us_o = timeit(lambda: ops.gemm(a, w, b, ops.TR_EPI_BF16, out=out))
        us_l = timeit(lambda: torch.nn.functional.linear(a, w, bb))
        us_m = timeit(lambda: torch.mm(a, w.t(), out=out))
        f = 2.0 * M * N * K / 1e6
        tot_l += min(us_l, us_m) * 3
        tot_o += us_o * 3
        print(f"tokens {tokens:3d} {name:4s} M={M:6d} N={N:4d} K={K:4d}: ours {us_o:7.2f} us {f / us_o:6.1f} TF | library linear {us_l:7.2f} us {f / us_l:6.1f} TF"
              f" | library mm (no bias) {us_m:7.2f} us {f / us_m:6.1f} TF")
print(f"sum over the forward's 48 Linear launches (3 blocks per token count): ours {tot_o:.0f} us, library {tot_l:.0f} us")
