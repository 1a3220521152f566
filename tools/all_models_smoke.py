#!/usr/bin/env python3
"""Forward every registered factory name once on the GPU (B=4, random weights): shapes, finiteness, token schedule."""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tokenreduction_amd as tra  # noqa: E402

args = types.SimpleNamespace(keep_rate=[0.7], reduction_loc=[3, 6, 9], dyvit_distill=False, k_neighbors=5, equal_weight=False,
                             cluster_iters=3, sinkhorn_eps=1.0, heuristic_pattern="l2", not_contiguous=False, min_radius=None)
x = torch.randn(4, 3, 224, 224).cuda()
bad = 0
for name in tra.list_models():
    try:
        torch.manual_seed(0)
        m = tra.create_model(name, pretrained=False, num_classes=1000, img_size=224, args=args).cuda().eval()
        with torch.no_grad():
            for blk in m.blocks:
                blk.attn.qkv.weight.mul_(4.0)
        out = m(x)
        logits = out[0] if isinstance(out, tuple) else out
        ok = logits.shape == (4, 1000) and bool(torch.isfinite(logits).all())
        print(f"{name:36s} {'ok ' if ok else 'BAD'} tokens {m._last_tokens}")
        bad += not ok
    except Exception as e:   # noqa: BLE001
        bad += 1
        print(f"{name:36s} FAILED {type(e).__name__}: {str(e)[:150]}")
print("ALL OK" if not bad else f"{bad} FAILED")
