"""Lab: key-blocked attention backward (tr_attention_bwd_long_bf16) against the register-resident one at the training shapes."""
import sys
import torch
sys.path.insert(0, ".")
from tokenreduction_amd import ops

for B, H, N in ((256, 6, 197), (256, 6, 138), (256, 6, 97), (256, 6, 68), (128, 12, 197), (64, 12, 577)):
    qkv = (torch.randn(B * N, 3 * H * 64, device="cuda") * 1.5).bfloat16()
    do = torch.randn(B * N, H * 64, device="cuda").bfloat16()
    res = {}
    for name, fn in (("short", ops.attention_bwd), ("long", ops.attention_bwd_long)):
        if name == "short" and N > 224:
            continue
        for _ in range(2):
            fn(qkv, do, B, N, H)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn(qkv, do, B, N, H)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 100
    print(f"B={B} H={H} N={N}: " + "  ".join(f"{k} {v:7.1f} us ({10.0 * B * H * N * N * 64 / v / 1e6:5.0f} TF useful)" for k, v in res.items()))
