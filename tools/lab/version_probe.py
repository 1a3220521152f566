import sys
sys.path.insert(0, "/root/repo")
import torch, bench
for fused in (True, False):
    model = bench.build_model(bench.MODEL, [0.7], [3,6,9], "cuda").train()
    x = torch.randn(32, 3, 224, 224, device="cuda"); y = torch.randint(0, 1000, (32,), device="cuda")
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.05, fused=fused)
    p = model.blocks[0].attn.qkv.weight
    losses = []
    for i in range(4):
        v0 = p._version
        k0 = model._pack()["key"][:3]
        out = model(x); loss = torch.nn.functional.cross_entropy(out, y)
        opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
        losses.append(round(loss.item(), 4))
        print("fused", fused, "step", i, "version", v0, "->", p._version, "repacked:", model._pack()["key"][:3] != k0)
    print("losses", losses)
