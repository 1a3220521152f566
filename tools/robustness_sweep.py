import sys, types, torch
sys.path.insert(0, ".")
import oracle, tokenreduction_amd as tra
from tests._params import make_params, make_images
from oracle import VitConfig
ok = True
for (fam, cls, img, patch, chans, D, H, depth, B, kr, loc) in [
    ("topk", "TopKVisionTransformer", 224, 16, 3, 128, 2, 3, 5, [0.6], [1]),      # keep = int(r * 196): the reference hard-codes 196 patches (topk.py:56)
    ("topk", "TopKVisionTransformer", 224, 16, 3, 256, 4, 2, 3, [0.3], [0]),      # D = 256: one 64-lane chunk per LayerNorm row
    ("evit", "EfficientVisionTransformer", 224, 16, 3, 1024, 16, 2, 2, [0.5], [1]),  # D = 1024: four chunks per row
    ("deit", "VisionTransformer", 224, 16, 1, 64, 1, 2, 1, [1.0], []),
    ("deit", "VisionTransformer", 96, 16, 3, 192, 3, 2, 77, [1.0], []),
    ("evit", "EfficientVisionTransformer", 160, 16, 3, 128, 2, 4, 3, [0.5], [0, 2]),
    ("tome", "ToMeVisionTransformer", 128, 16, 3, 128, 2, 4, 2, [0.6], [0, 1, 2, 3]),
]:
    cfg = VitConfig(family=fam, img_size=img, patch_size=patch, in_chans=chans, num_classes=12, embed_dim=D, depth=depth, num_heads=H,
                    keep_rate=kr, reduction_loc=loc)
    params = make_params(cfg, 5, 4.0)
    args = types.SimpleNamespace(keep_rate=kr, reduction_loc=loc)
    try:
        m = getattr(tra, cls)(img_size=img, patch_size=patch, in_chans=chans, embed_dim=D, depth=depth, num_heads=H, mlp_ratio=4,
                              qkv_bias=True, num_classes=12, args=args)
        m.load_state_dict(params, strict=True)
        m = m.cuda().eval()
        x = make_images(B, img, 9, chans)
        got = m(x.cuda()).cpu()
        want = oracle.forward(params, x, cfg, precision="bf16")
        rel = ((got - want).norm() / want.norm()).item()
        print(f"{fam:5s} img{img} p{patch} c{chans} D{D} B{B}: tokens {m._last_tokens} rel L2 vs oracle_bf16 {rel:.2e}")
        ok &= rel < 0.3
    except Exception as e:
        ok = False
        print(f"{fam} img{img} p{patch}: FAILED {type(e).__name__}: {e}")
print("ALL OK" if ok else "SOME FAILED")
