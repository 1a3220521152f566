#!/usr/bin/env python3
"""HBM traffic of every token-reduction kernel against its ALGORITHMIC bytes (every input read once, every output
written once), from the PMC passes of tools/prof_r04.sh:

    python tools/traffic_table.py <tag>          reads profiles/<tag>_<workload>_pmc_traffic.json, prints a markdown table

Shapes per workload (token counts entering / leaving each reduction stage) are the static token plan of the model the workload runs;
a kernel launched once per stage is compared with the MEAN over the stages, as the PMC summary averages over its launches.
"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# workload -> (B, D, H, [(tokens in, tokens out)] per reduction stage)
S_IN = [(197, 138), (138, 97), (97, 68)]
B_IN = [(197, 99), (99, 50), (50, 25)]
B384 = [(577, 145), (145, 37), (37, 10)]
TOME = [(197 - 16 * i, 197 - 16 * (i + 1)) for i in range(11)] + [(21, 11)]
WL = {
    "headline": (256, 384, 6, S_IN), "evit_small": (256, 384, 6, [(197, 139), (139, 98), (98, 69)]), "tome": (256, 384, 6, TOME),
    "ats_small": (256, 384, 6, S_IN), "dpcknn_small": (256, 384, 6, S_IN), "sit_small": (256, 384, 6, S_IN),
    "atsb_train": (128, 768, 12, B_IN), "dpcknnb_train": (128, 768, 12, B_IN),
    "kmedb384": (64, 768, 12, B384), "sinkb384": (64, 768, 12, B384),
}


def mean(f, stages):
    return sum(f(n, k) for n, k in stages) / len(stages)


def algorithmic(kernel, B, D, H, st, wl=""):
    """Bytes per launch, or None when the kernel is not a reduction kernel of this table.  n = tokens entering the stage, k = leaving it
    (CLS included in both); patch tokens p = n - 1, clusters / kept patches c = k - 1."""
    K = kernel
    if K.startswith("cls_topk_kernel"):          # CLS attention rows [B,H,n] fp32 -> kept ids (+ the complement for EViT), int32
        if "topk" not in wl and "evit" not in wl and wl != "headline":
            return None                          # (the clustering families use it as a plain top-K over one score row per image)
        return mean(lambda n, k: B * (4 * H * n + 4 * (n - 1)), st)
    if K.startswith("gather_layernorm_kernel"):  # kept rows: x fp32 in + bf16 delta in, x fp32 out + bf16 y out
        if wl.startswith("evit"):                # EViT also READS the rows it drops (their weighted mean is the fused token, evit.py:117-124)
            return mean(lambda n, k: B * (n * D * 6 + k * D * 6), st)
        return mean(lambda n, k: B * k * D * 12, st)
    if K.startswith("ats_sample_kernel"):        # CLS rows + the V third of qkv (value norms, ats.py:60-66) -> ids and masks
        return mean(lambda n, k: B * (4 * H * n + 2 * n * H * 64 + 8 * k), st)
    if K.startswith("ats_gather_kernel"):        # sampled rows of the stream (fp32) + of the attention output (bf16), in and out
        return mean(lambda n, k: B * k * D * (4 + 2) * 2, st)
    if K.startswith("ats_scatter_kernel"):       # backward of the gather: sampled rows in, the same rows out (the zero fill of the rest is a memset)
        return mean(lambda n, k: B * 2 * k * D * (4 + 2), st)
    if K.startswith("dpcknn_fused_kernel"):      # patch tokens fp32 + noise in; assignment + centres out
        return mean(lambda n, k: B * ((n - 1) * D * 4 + 8 * (n - 1) + 4 * (k - 1)), st)
    if K.startswith("sqnorm"):
        return mean(lambda n, k: B * (n - 1) * (4 * D + 4), st)
    if K.startswith("dist_mfma_kernel"):         # staged path: tokens in, distance matrix out
        return mean(lambda n, k: B * ((n - 1) * D * 4 + 4 * (n - 1) ** 2), st)
    if K.startswith("density_kernel") or K.startswith("parent_score_kernel"):
        return mean(lambda n, k: B * (4 * (n - 1) ** 2 + 8 * (n - 1)), st)
    if K.startswith("assign_kernel"):
        return mean(lambda n, k: B * (4 * (n - 1) * (k - 1) + 4 * (n - 1)), st)      # the centre columns of the distance matrix
    if K.startswith("token_weight_kernel"):      # score Linear(D -> 1) + exp (dpcknn.py:257)
        return mean(lambda n, k: B * (n - 1) * (4 * D + 4), st)
    if K.startswith("cluster_merge_layernorm_kernel"):
        return mean(lambda n, k: B * (n * D * 4 + 8 * n + k * D * 6), st)
    if K.startswith("cluster_merge_bwd_kernel"):
        return mean(lambda n, k: B * (k * D * 4 + n * D * 4 + n * D * (4 + 2) + 8 * n), st)
    if K.startswith("tome_match_kernel"):        # the K third of qkv, bf16 -> (unm, src, dst) ids
        return mean(lambda n, k: B * (2 * n * H * 64 + 4 * n), st)
    if K.startswith("tome_merge_layernorm_kernel"):
        return mean(lambda n, k: B * (n * D * 6 + 8 * n + k * D * 6 + 4 * k), st)
    if K.startswith("kmed_iterate_kernel") or K.startswith("kmed_rowcost_kernel"):
        return mean(lambda n, k: B * (4 * (n - 1) ** 2 + 4 * (k - 1)), st)
    if K.startswith("kmed_weight_kernel"):
        return mean(lambda n, k: B * (4 * H * n + 4 * n), st)
    if K.startswith("attention_colsum_kernel"):  # 384^2 only: column sums of softmax(qk^T): q and k thirds of qkv in, [B,H,4,n] out
        return B * (2 * 2 * st[0][0] * H * 64 + 16 * H * st[0][0])
    if K.startswith("sinkhorn_global_kernel") or K.startswith("sinkhorn_regs_kernel"):   # first stage at 384^2: the [p, c] scores in, the plan out
        return B * (st[0][0] - 1) * (st[0][1] - 1) * 8
    if K.startswith("sinkhorn_kernel"):
        return mean(lambda n, k: B * (n - 1) * (k - 1) * 8, st[1:])
    if K.startswith("softmerge_mfma_kernel"):    # weights [B,p,c] fp32 + tokens fp32 in, merged [B,c,D] fp32 out
        return mean(lambda n, k: B * (4 * (n - 1) * (k - 1) + 4 * (n - 1) * D + 4 * (k - 1) * D), st)
    if K.startswith("token_softmax_kernel"):
        return mean(lambda n, k: B * (n - 1) * (k - 1) * 8, st)
    if K.startswith("rownorm_kernel"):
        return mean(lambda n, k: B * (n - 1) * D * (4 + 4 + 2), st)
    if K.startswith("mlp_fused_kernel") and wl == "headline":     # fused eval Mlp: LN2 output in, fc2 output out (bf16), the packed weights once
        wts = 2 * D * 4 * D * 2
        if "true>" in K.split("(")[0] and "<false, true>" in K:
            # round 6: the norm2 inside the launch (one-round launches: blocks 7, 8 at 97 tokens per image): fp32 stream row + bf16 residual in
            m = B * 97
            return m * D * (4 + 2 + 2) + wts
        if "<false, false>" in K:
            # round 6: the plain launch serves blocks 0..6 (197 x 3, 138 x 3, 97 x 1 tokens per image)
            ms = [B * 197] * 3 + [B * 138] * 3 + [B * 97]
            return sum(m * D * 2 * 2 + wts for m in ms) / len(ms)
        # round 5: nine launches (197 x 3, 138 x 3, 97 x 3 tokens per image)
        ms = [B * 197, B * 138, B * 97]
        return sum(m * D * 2 * 2 + wts for m in ms) / len(ms)
    return None


def main():
    tag = sys.argv[1]
    sq_tag = sys.argv[2] if len(sys.argv) > 2 else None      # profiles/<sq_tag>_pmc_sq_<workload>.json (tools/prof_r05_sq.sh): busy-unit columns
    print("| workload | kernel | µs | algorithmic MB | HBM MB (fetch + write) | ratio | GB/s |" + (" MFMA busy | LDS busy | VALU issue | waiting on an instruction |" if sq_tag else ""))
    print("|---|---|---|---|---|---|---|" + ("---|---|---|---|" if sq_tag else ""))
    over = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{tag}_*_pmc_traffic.json"))):
        w = os.path.basename(f)[len(tag) + 1:-len("_pmc_traffic.json")]
        if w not in WL:
            continue
        B, D, H, st = WL[w]
        for k, v in json.load(open(f)).items():
            if k.startswith("_"):
                continue
            a = algorithmic(k, B, D, H, st, w)
            if a is None:
                continue
            m = v["hbm_bytes_per_launch"]
            r = m / a
            name = k.split("(")[0]
            sq = ""
            if sq_tag:
                sf = os.path.join(ROOT, "profiles", f"{sq_tag}_pmc_sq_{w}.json")
                rec = json.load(open(sf)).get(name) if os.path.exists(sf) else None
                # busy fractions at the nominal 2.4 GHz (round 5's files carry them under the clock-dependent names only)
                if rec:
                    mf = rec.get("mfma_busy_frac_at_2p4ghz", rec.get("mfma_busy_frac")) or 0.0
                    ld = rec.get("lds_busy_frac_at_2p4ghz", rec.get("lds_busy_frac")) or 0.0
                sq = (f" {100 * mf:.0f} % | {100 * ld:.0f} % | {100 * rec['valu_inst_frac']:.0f} % | "
                      f"{100 * rec['wait_inst_frac']:.0f} % |") if rec else " | | | |"
            print(f"| {w} | `{name}` | {v['avg_us']:.1f} | {a / 1e6:.1f} | {m / 1e6:.1f} ({v['fetch_bytes_per_launch'] / 1e6:.1f} + "
                  f"{v['write_bytes_per_launch'] / 1e6:.1f}) | {r:.2f} | {v['hbm_gbps']:.0f} |" + sq)
            if r > 1.5:
                over.append((w, name, r))
    print()
    print("Above 1.5 x algorithmic:", ", ".join(f"`{n}` ({w}: {r:.1f} x)" for w, n, r in over) or "none")


if __name__ == "__main__":
    main()
