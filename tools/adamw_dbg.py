import sys, torch
sys.path.insert(0, "/root/repo")
from tests._params import GOLDEN_CASES, make_images, grad_labels
from tests.test_hip_model import build_model
from tokenreduction_amd.optim import FusedAdamW
case = GOLDEN_CASES["topk_micro"]
x = make_images(case["batch"], 224, case["xseed"]).cuda(); y = grad_labels(case).cuda()
def run(kind, steps, lrs):
    model, _, _ = build_model(case); model.viz_mode = False; model.train()
    ps = list(model.parameters())
    opt = torch.optim.AdamW(ps, lr=2e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.05, fused=True) if kind == "torch" else FusedAdamW(ps, lr=2e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.05, model=model)
    snaps = []
    for it in range(steps):
        for g in opt.param_groups: g["lr"] = lrs[it]
        loss = torch.nn.functional.cross_entropy(model(x), y)
        opt.zero_grad(set_to_none=True); loss.backward()
        grads = [p.grad.detach().clone() for p in ps]
        opt.step()
        snaps.append((grads, [p.detach().clone() for p in ps], [opt.state[p]["exp_avg"].clone() for p in ps], [opt.state[p]["exp_avg_sq"].clone() for p in ps]))
    return snaps, [n for n, _ in model.named_parameters()]
for lrs in ([2e-3, 2e-3, 2e-3],):
    a, names = run("torch", 3, lrs); b, _ = run("hip", 3, lrs)
    print("lrs", lrs)
    import numpy as np
    cat = lambda it, k: torch.cat([t.flatten() for t in a[it][k]])
    catb = lambda it, k: torch.cat([t.flatten() for t in b[it][k]])
    np.savez("gpurun_out/adamw_cases.npz", g1=cat(1, 0).cpu().numpy(), p0=cat(0, 1).cpu().numpy(), m0=cat(0, 2).cpu().numpy(), v0=cat(0, 3).cpu().numpy(),
             p1=cat(1, 1).cpu().numpy(), m1=cat(1, 2).cpu().numpy(), v1=cat(1, 3).cpu().numpy(), g0=cat(0, 0).cpu().numpy(),
             hp1=catb(1, 1).cpu().numpy(), hm1=catb(1, 2).cpu().numpy(), hv1=catb(1, 3).cpu().numpy())
    for it in range(3):
        tot = [0, 0, 0, 0]; worst = None
        for i, n in enumerate(names):
            for k in range(4):
                d = int((a[it][k][i] != b[it][k][i]).sum())
                tot[k] += d
                if k == 1 and d and worst is None: worst = (n, d, a[it][1][i].numel())
        if it == 1:
            for k, nm in ((2, "exp_avg"), (3, "exp_avg_sq")):
                shown = 0
                for i, n in enumerate(names):
                    mm = (a[it][k][i] != b[it][k][i]).flatten().nonzero().flatten()[:3]
                    for j in mm.tolist():
                        if shown < 6:
                            print(f"     {nm} {n}[{j}]: torch {a[it][k][i].flatten()[j].item():.9e} hip {b[it][k][i].flatten()[j].item():.9e}  grad {a[it][0][i].flatten()[j].item():.6e} prev {a[it-1][k][i].flatten()[j].item():.9e}")
                            shown += 1
        print(f"  step {it}: mismatching elements grad {tot[0]} param {tot[1]} exp_avg {tot[2]} exp_avg_sq {tot[3]}  first: {worst}")
