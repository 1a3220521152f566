#!/bin/bash
# SQ counter passes over the isolated attention forward (tools/attn_lab.py): bash tools/prof_attn_sq.sh <tag> [N]
TAG=${1:-attn}; N=${2:-197}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${TAG}_sq1 -o sq -- python3 tools/attn_lab.py $N > gpurun_out/${TAG}_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/${TAG}_sq2 -o sq -- python3 tools/attn_lab.py $N > gpurun_out/${TAG}_sq2.log 2>&1
python3 - <<PY
import csv, collections, glob
for d in ("gpurun_out/${TAG}_sq1", "gpurun_out/${TAG}_sq2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print("no csv in", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60]
        if "attention" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, c in acc.items():
        print(k, "launches", len(n[k]))
        for cn, v in sorted(c.items()): print(f"   {cn:28s} {v / len(n[k]):16.0f} per launch")
PY
