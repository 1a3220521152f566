#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/) into the committed summaries under profiles/.

  python tools/prof_summary.py <tag> <kernel_stats.csv> <pmc_fetch counter_collection.csv> <pmc_write counter_collection.csv> [outdir]

Writes profiles/<tag>_kernel_stats.csv (verbatim copy of rocprofv3 --kernel-trace --stats), and
profiles/<tag>_pmc_traffic.json: per kernel, HBM-side bytes per launch from the TCC counters, collected in SEPARATE
--pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass) and corrected as MI355X_MICROARCH.md section HBM prescribes:
counter unit = KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B -> doubled;
WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import collections
import csv
import json
import re
import shutil
import sys

tag, stats, fetch, write = sys.argv[1:5]
outdir = sys.argv[5] if len(sys.argv) > 5 else "profiles"      # on the GPU box: a directory under gpurun_out/ (merged back by gpurun)
shutil.copy(stats, f"{outdir}/{tag}_kernel_stats.csv")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[short(r["Kernel_Name"])]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return acc


f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
# average duration per launch from the (un-countered) kernel-stats run of the same command: GB/s = counter bytes / that time
avg_ns = {}
for r in csv.DictReader(open(stats)):
    avg_ns[short(r["Name"])] = float(r["AverageNs"])
out = {}
for k in sorted(set(f) | set(w)):
    if k.startswith(("at::", "void at::", "__amd_rocclr")) or "at::native" in k:      # framework kernels of the harness (fills, casts), not ours
        continue
    fb = 2.0 * 1024.0 * f[k][0] / max(f[k][1], 1)      # gfx950: x2
    wb = 1024.0 * w[k][0] / max(w[k][1], 1)
    out[k] = dict(launches_profiled=f[k][1], fetch_bytes_per_launch=round(fb), write_bytes_per_launch=round(wb),
                  hbm_bytes_per_launch=round(fb + wb))
    if k in avg_ns:
        out[k]["avg_us"] = round(avg_ns[k] / 1e3, 2)
        out[k]["hbm_gbps"] = round((fb + wb) / avg_ns[k], 1)
sys.path.insert(0, ".")
import bench  # noqa: E402
out["_kernel_source_hash"] = bench.kernel_source_hash()          # bench.py only trusts this file for the sources it was collected on
json.dump(out, open(f"{outdir}/{tag}_pmc_traffic.json", "w"), indent=1)
out.pop("_kernel_source_hash")
for k, v in out.items():
    print(f"{k[:60]:60s} fetch {v['fetch_bytes_per_launch']/1e6:9.2f} MB  write {v['write_bytes_per_launch']/1e6:9.2f} MB  {v.get('avg_us', 0):8.1f} us  "
          f"{v.get('hbm_gbps', 0):7.0f} GB/s  ({v['launches_profiled']} launches)")
