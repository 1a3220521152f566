"""Lab: the four parameter-gradient products of a block as four launches, two pairs, or one group of four (tr_linear_bwd_group), same process."""
import sys
import torch
sys.path.insert(0, ".")
from tokenreduction_amd import ops

for label, M, D, Hd in (("DeiT-S 197 tok B=256", 50432, 384, 1536), ("DeiT-S 97 tok", 24832, 384, 1536), ("DeiT-B 197 tok B=128", 25216, 768, 3072)):
    mk = lambda n: torch.randn(M, n, device="cuda").bfloat16()
    layers = [(mk(D), mk(Hd)), (mk(Hd), mk(D)), (mk(D), mk(D)), (mk(3 * D), mk(D))]       # fc2, fc1, proj, qkv
    outs = [(torch.empty(dy.shape[1], x.shape[1], device="cuda"), torch.empty(dy.shape[1], device="cuda")) for dy, x in layers]

    def singles():
        for (dy, x), (dw, db) in zip(layers, outs):
            ops.linear_bwd_params(dy, x, dw=dw, db=db)

    def pairs():
        ops.linear_bwd_group(layers[:2], outs=outs[:2])
        ops.linear_bwd_group(layers[2:], outs=outs[2:])

    def four():
        ops.linear_bwd_group(layers, outs=outs)
    res = {}
    for rep in range(3):
        for name, fn in (("singles", singles), ("pairs", pairs), ("four", four)):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name] = min(res.get(name, 1e9), e0.elapsed_time(e1) * 100)
    fl = 2.0 * M * (D * Hd * 2 + D * D + 3 * D * D)
    print(f"{label}: four launches {res['singles']:7.1f} us, two pairs {res['pairs']:7.1f} us, one group {res['four']:7.1f} us  "
          f"({fl / res['singles'] / 1e6:.0f} / {fl / res['pairs'] / 1e6:.0f} / {fl / res['four'] / 1e6:.0f} TFLOP/s)")
