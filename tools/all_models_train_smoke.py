#!/usr/bin/env python3
"""One training step (forward, cross-entropy, backward) of every registered factory name on the GPU (B=4, random weights): finite loss,
a finite gradient for every parameter, nothing raised (no whitelist: since round 3 DyViT / SiT train at DeiT-T width too, their 96-wide
hidden layers zero-padded to 128 through the tape and the backward)."""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tokenreduction_amd as tra  # noqa: E402

x = torch.randn(4, 3, 224, 224).cuda()
y = torch.randint(0, 1000, (4,)).cuda()
bad = 0
for name in tra.list_models():
    args = types.SimpleNamespace(keep_rate=[0.7], reduction_loc=[3, 6, 9], dyvit_distill=False, k_neighbors=5, equal_weight=False,
                                 cluster_iters=3, sinkhorn_eps=1.0, heuristic_pattern="l2", not_contiguous=False, min_radius=None)
    try:
        torch.manual_seed(0)
        m = tra.create_model(name, pretrained=False, num_classes=1000, img_size=224, args=args).cuda().train()
        m.viz_mode = False
        out = m(x)
        logits = out[0] if isinstance(out, tuple) else out
        loss = torch.nn.functional.cross_entropy(logits, y)
        if isinstance(out, tuple) and len(out) == 2 and isinstance(out[1], (list, tuple)):       # DyViT: keep every output in the graph
            loss = loss + sum(((s.mean(1) - 0.5) ** 2).mean() for s in out[1])
        if "teacher" in name:                    # inference-only by construction (run under no_grad by the DyViT loss): no graph
            assert not logits.requires_grad and out[1].shape == (4, 196, m.embed_dim)
            print(f"{name:36s} ok  (teacher: inference executor in any mode) tokens {m._last_tokens}")
            continue
        loss.backward()
        torch.cuda.synchronize()
        missing = [n for n, p in m.named_parameters() if p.requires_grad and (p.grad is None or not bool(torch.isfinite(p.grad).all()))]
        ok = bool(torch.isfinite(loss)) and not missing
        print(f"{name:36s} {'ok ' if ok else 'BAD'} loss {loss.item():.4f} tokens {m._last_tokens} {missing[:3]}")
        bad += not ok
    except Exception as e:   # noqa: BLE001
        print(f"{name:36s} FAILED {type(e).__name__}: {str(e)[:150]}")
        bad += 1
print("ALL OK" if not bad else f"{bad} FAILED")
