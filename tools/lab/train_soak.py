"""Lab: race screen of the TRAINING path -- the same forward + loss + backward (same weights, same batch, no optimizer step) n times back to
back; every gradient must equal the first run's bit for bit (the backward kernels sum in a fixed order -- except DyViT's predictor head, whose two-class Linear accumulates
its weight gradient with LDS float atomics: differences at the 1e-6 level there, nowhere else -- so a difference is a race or an uninitialised read).  tools/lab/train_soak.py [n] [family ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
args = sys.argv[1:]
n = int(args[0]) if args else 60
fams = args[1:] or ["topk_small_patch16_224", "evit_small_patch16_224", "tome_small_patch16_224", "dyvit_small_patch16_224", "ats_base_patch16_224",
                    "dpcknn_base_patch16_224", "sinkhorn_small_patch16_224", "sit_small_patch16_224", "patchmerger_small_patch16_224", "kmedoids_small_patch16_224",
                    "heuristic_small_patch16_224", "deit_small_patch16_224_local"]
torch.cuda.set_device(0)
for name in fams:
    B = 64 if "base" in name else 128
    tome = name.startswith("tome")
    kr = [196 - 16 * (i + 1) for i in range(12)] if tome else ([0.5] if "base" in name else [0.7])       # ToMe: r = 16 in every block (configs[2])
    try:
        model = bench.build_model(name, kr, list(range(12)) if tome else [3, 6, 9], "cuda").train()
    except Exception as e:                                                 # a family the factory does not know under this name
        print(f"{name}: skipped ({type(e).__name__}: {str(e)[:80]})", flush=True)
        continue
    g0 = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, 3, 224, 224, device="cuda", generator=g0)
    y = torch.randint(0, 1000, (B,), device="cuda", generator=g0)
    ref, bad, worst = None, {}, 0.0
    for i in range(n):
        if hasattr(model, "_noise_slot"):
            pass
        torch.manual_seed(7)                                               # gumbel / dropout noise of the families that draw any: the same every run
        torch.cuda.manual_seed(7)
        out = model(x)
        loss = torch.nn.functional.cross_entropy(out[0] if isinstance(out, tuple) else out, y)
        for p in model.parameters():
            p.grad = None
        loss.backward()
        grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        if ref is None:
            ref = grads
            continue
        for k, gk in grads.items():
            if not torch.equal(gk, ref[k]):
                bad[k] = bad.get(k, 0) + 1
                worst = max(worst, float((gk - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30)))
    torch.cuda.synchronize()
    print(f"{name} B={B}: {n} steps, loss {float(loss.detach()):.5f}; parameters whose gradient ever differed from the first run: {len(bad)} of {len(ref)}"
          + (f" (worst relative difference {worst:.3g}; e.g. {sorted(bad.items(), key=lambda kv: -kv[1])[:4]})" if bad else ""), flush=True)
    del model
