#!/usr/bin/env python3
"""Which roundings to bf16 cost the logits their 1e-3?  

The bf16 product path rounds at seven kinds of site per block; the bf16x3 mode (split-bf16 operands, fp32 activations) removes all of
them at 3 MFMAs per product and fp32 activation traffic.  This tool measures, on the CPU oracle (the restatement of the reference, fp32
torch), how far the logits of a dense DeiT-S move from the all-fp32 forward when

  * exactly ONE kind of site is rounded to bf16 and everything else stays fp32 ("only"), and
  * everything is rounded EXCEPT one kind of site ("all but"),

for the plain initialisation (trunc_normal 0.02: near-uniform attention, the well-conditioned case) and for the benchmark's qkv x 4
model.  Sites (per block unless noted), each = operand AND stored-output rounding as the bf16 executor does it:
  patch  PatchEmbed projection (once)        ln    LayerNorm outputs xn1 / xn2 (stored bf16)
  qkv    qkv Linear: W and the stored q,k,v   attn  softmax numerators P -> bf16 for P.V, stored attention output -> bf16
  proj   proj Linear: W and its stored output fc1   fc1: W, stored GELU output
  fc2    fc2 Linear: W and its stored output
A dense model is used so that no discrete token decision amplifies the differences.  Output: a markdown table (profiles/r03_precision_ablation.md).

    python tools/precision_ablation.py [batch]
"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import oracle  # noqa: E402  (CPU restatement of the reference: used as the measuring instrument of a lab tool)
from oracle.vit import _r, gelu_erf, layer_norm, patch_embed, embed_tokens, head  # noqa: E402

SITES = ("patch", "ln", "qkv", "attn", "proj", "fc1", "fc2")


def forward(p, x, cfg, P):
    """Dense DeiT forward (deit_viz.py:186-212) with a precision per site kind."""
    tok = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg.patch_size, P["patch"])
    h = embed_tokens(tok, p["cls_token"], p["pos_embed"])
    H = cfg.num_heads
    for i in range(cfg.depth):
        pre = f"blocks.{i}."
        xn = layer_norm(h, p[pre + "norm1.weight"], p[pre + "norm1.bias"], cfg.ln_eps, P["ln"])
        B, N, D = xn.shape
        qkv = _r(_r(xn, P["qkv"]) @ _r(p[pre + "attn.qkv.weight"], P["qkv"]).t() + p[pre + "attn.qkv.bias"], P["qkv"])
        q, k, v = qkv.reshape(B, N, 3, H, D // H).permute(2, 0, 3, 1, 4)
        s = (q @ k.transpose(-2, -1)) * (D // H) ** -0.5
        m = s.amax(-1, keepdim=True)
        e = torch.exp(s - m)
        o = (_r(e, P["attn"]) @ v) / e.sum(-1, keepdim=True)
        o = _r(o.transpose(1, 2).reshape(B, N, D), P["attn"])
        h = h + _r(_r(o, P["proj"]) @ _r(p[pre + "attn.proj.weight"], P["proj"]).t() + p[pre + "attn.proj.bias"], P["proj"])
        xn2 = layer_norm(h, p[pre + "norm2.weight"], p[pre + "norm2.bias"], cfg.ln_eps, P["ln"])
        hid = _r(gelu_erf(_r(xn2, P["fc1"]) @ _r(p[pre + "mlp.fc1.weight"], P["fc1"]).t() + p[pre + "mlp.fc1.bias"]), P["fc1"])
        h = h + _r(_r(hid, P["fc2"]) @ _r(p[pre + "mlp.fc2.weight"], P["fc2"]).t() + p[pre + "mlp.fc2.bias"], P["fc2"])
    return head(h, p["norm.weight"], p["norm.bias"], p["head.weight"], p["head.bias"], cfg.ln_eps, "fp32")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    from tests._params import make_images, make_params
    cfg = oracle.VitConfig(family="deit", embed_dim=384, depth=12, num_heads=6, num_classes=1000, keep_rate=[1.0], reduction_loc=[])
    shape = types.SimpleNamespace(embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, num_classes=1000, img_size=224, patch_size=16, in_chans=3)
    x = make_images(B, 224, 4242)
    rows = []
    with torch.no_grad():
        for label, gain in (("plain init", 1.0), ("qkv x 4", 4.0)):
            p = make_params(shape, 81, gain)
            ref = forward(p, x, cfg, {s: "fp32" for s in SITES})
            res = {}
            for s in SITES + ("ALL",):
                only = {t: ("bf16" if (t == s or s == "ALL") else "fp32") for t in SITES}
                res[("only", s)] = (forward(p, x, cfg, only) - ref).abs().max().item()
                if s != "ALL":
                    allbut = {t: ("fp32" if t == s else "bf16") for t in SITES}
                    res[("allbut", s)] = (forward(p, x, cfg, allbut) - ref).abs().max().item()
            rows.append((label, ref.abs().max().item(), res))
    out = ["# bf16 rounding sites vs the 1e-3 logit tolerance (CPU oracle, dense DeiT-S, batch %d; tools/precision_ablation.py)" % B, "",
           "max |logit - all-fp32 logit| over the batch; north_star's tolerance is 1e-3.", ""]
    for label, scale, res in rows:
        out += [f"## {label} (largest |logit| {scale:.2f})", "", "| site | only this site in bf16 | everything in bf16 but this site |", "|---|---|---|"]
        for s in SITES:
            out.append(f"| {s} | {res[('only', s)]:.2e} | {res[('allbut', s)]:.2e} |")
        out += [f"| all seven | {res[('only', 'ALL')]:.2e} | -- |", ""]
    text = "\n".join(out)
    print(text)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_precision_ablation.md"), "w") as f:
        f.write(text + "\n")


if __name__ == "__main__":
    main()
