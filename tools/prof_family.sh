#!/bin/bash
# usage: bash tools/prof_family.sh <model> <keep spec> [batch]   -> gpurun_out/prof_<model>/stats_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1 -o stats -- python3 tools/run_model.py $1 $2 ${3:-256} 10 > gpurun_out/prof_$1.log 2>&1
tail -1 gpurun_out/prof_$1.log
python3 - <<PY
import csv, re
rows = list(csv.DictReader(open("gpurun_out/prof_$1/stats_kernel_stats.csv")))
for r in rows[:16]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    print(f"{n[:52]:52s} calls {r['Calls']:>5s}  per_fwd_us {float(r['TotalDurationNs']) / 1e3 / 13:8.1f}  avg_us {float(r['AverageNs']) / 1e3:7.1f}  {r['Percentage']}%")
PY
