#!/bin/bash
# Training-step kernel stats of every family (DeiT-S width, batch 256): which family-specific backward kernels matter?
#   bash tools/prof_families_train.sh      (on the GPU box, through gpurun); summaries in gpurun_out/train_fam_<name>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in ats_small dpcknn_small kmedoids_small sinkhorn_small sit_small patchmerger_small dyvit_small evit_small tome_small heuristic_small; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/train_fam_$m -o stats -- python3 tools/train_step.py ${m}_patch16_224 256 4 > gpurun_out/train_fam_$m.log 2>&1
  python3 - $m <<'PY'
import csv, sys, glob
m = sys.argv[1]
f = glob.glob(f"gpurun_out/train_fam_{m}/**/stats_kernel_stats.csv", recursive=True)
if not f:
    print(m, "no stats"); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
out = [f"{m}: {tot / 6e6:.2f} ms of kernels per step (6 steps)"]
for r in rows[:40]:
    n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:58]
    if any(k in n for k in ("gemm_bf16_pc", "wgrad_pc", "ln_bwd", "layernorm_half", "attention_bwd_kernel", "attention_kernel", "multi_tensor", "partial_reduce", "elementwise", "im2col", "gemm_bf16_persistent")):
        continue
    out.append(f"   {n:58s} {int(r['Calls']):5d} x {float(r['AverageNs']) / 1e3:8.1f} us = {100 * int(r['TotalDurationNs']) / tot:5.1f} %")
open(f"gpurun_out/train_fam_{m}.txt", "w").write("\n".join(out[:12]) + "\n")
print("\n".join(out[:9]))
PY
  rm -rf gpurun_out/train_fam_$m
done
