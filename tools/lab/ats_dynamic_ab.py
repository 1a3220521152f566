import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda")
for name, bsz in (("ats_small_patch16_224", 256), ("ats_base_patch16_224", 128)):
    for kr in ([0.7], [0.5]):
        m = bench.build_model(name, kr, [3, 6, 9], dev)
        x = torch.randn(bsz, 3, 224, 224, generator=torch.Generator().manual_seed(7)).to(dev)
        a = bench.quick_images_per_s(m, x); ta = list(m._last_tokens)
        m.use_graph = False
        a2 = bench.quick_images_per_s(m, x)
        m.use_graph = True
        m.dynamic_width = True
        b = bench.quick_images_per_s(m, x); tb = list(m._last_tokens)
        print(f"{name} kr{kr} B={bsz}: static {a:9.1f} img/s (plain launches {a2:9.1f}) tokens {ta[3]},{ta[6]},{ta[9]}   dynamic {b:9.1f} img/s tokens {tb[3]},{tb[6]},{tb[9]}", flush=True)
