"""Lab: fc1 of the training forward as one launch (tr_gemm_gelu_keep_bf16) against GEMM + elementwise GELU, same process."""
import sys
import torch
sys.path.insert(0, ".")
from tokenreduction_amd import ops

for label, M, N, K in (("DeiT-S 197 tok B=256", 50432, 1536, 384), ("DeiT-S 97 tok", 24832, 1536, 384), ("DeiT-B 197 tok B=128", 25216, 3072, 768),
                       ("DeiT-B 99 tok", 12672, 3072, 768)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.zeros(N, device="cuda")

    def two():
        return ops.gelu(ops.gemm(a, w, b, ops.TR_EPI_BF16))

    def one():
        return ops.gemm_gelu_keep(a, w, b)
    res = {}
    for rep in range(3):
        for name, fn in (("gemm+gelu", two), ("keep", one)):
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
    print(f"{label}: two launches {res['gemm+gelu']:7.1f} us, one {res['keep']:7.1f} us ({100 * (res['keep'] / res['gemm+gelu'] - 1):+.1f} %)")
