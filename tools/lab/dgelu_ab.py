"""Lab: fc2's data gradient with the GELU backward in the epilogue (tr_gemm_dgelu_bf16) against GEMM + elementwise pass, same process."""
import sys
import torch
sys.path.insert(0, ".")
from tokenreduction_amd import ops

for label, M, N, K in (("DeiT-S 197 tok B=256", 50432, 1536, 384), ("DeiT-S 97 tok", 24832, 1536, 384), ("DeiT-B 197 tok B=128", 25216, 3072, 768),
                       ("DeiT-B 99 tok", 12672, 3072, 768)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.zeros(N, device="cuda")
    pre = torch.randn(M, N, device="cuda").bfloat16()

    def two():
        return ops.gelu_bwd(pre, ops.gemm(a, w, b, ops.TR_EPI_BF16))

    def one():
        return ops.gemm_dgelu(a, w, pre)

    def plain():
        return ops.gemm(a, w, b, ops.TR_EPI_BF16)
    res = {}
    for rep in range(3):
        for name, fn in (("two", two), ("one", one), ("plain", plain)):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name] = min(res.get(name, 1e9), e0.elapsed_time(e1) * 50)
    print(f"{label}: GEMM alone {res['plain']:7.1f} us, GEMM + gelu_bwd {res['two']:7.1f} us, fused {res['one']:7.1f} us ({100 * (res['one'] / res['two'] - 1):+.1f} %)")
