#!/usr/bin/env python3
"""Condense a rocprofv3 SQ counter pass (tools/prof_sq.sh) into profiles/<tag>_pmc_sq.json: per kernel, averages per launch of
MFMA busy cycles, wave cycles and the wait split, plus the derived fractions.

  python tools/prof_sq_summary.py <tag> <counter_collection.csv>

SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's 1024 SIMDs (16 per v_mfma_f32_16x16x32_bf16, 32 per
32x32x16); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY count quad-cycles per wave (MI355X_MICROARCH.md, PMC notes)."""
import collections
import csv
import json
import re
import sys

tag, path = sys.argv[1:3]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
dur = collections.defaultdict(float)
for r in csv.DictReader(open(path)):
    k = short(r["Kernel_Name"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in cnt[k]:
        cnt[k].add(r["Dispatch_Id"])
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
for k, c in acc.items():
    if not k.startswith(("gemm", "attention", "layernorm", "gather", "cls_", "im2col")):
        continue
    n = len(cnt[k])
    us = dur[k] / n / 1e3
    mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n
    wave = c.get("SQ_WAVE_CYCLES", 0.0)
    row = dict(launches_profiled=n, avg_us_under_pmc=round(us, 1), mfma_busy_cycles_per_launch=round(mfma),
               # fraction of the launch during which a SIMD's matrix pipe is busy, at the 2.4 GHz the 2.5 PF peak is quoted on
               mfma_busy_frac_at_2p4ghz=round(mfma / (1024 * us * 1e-6 * 2.4e9), 4) if us else None)
    if wave:
        row.update(wait_any_frac=round(c.get("SQ_WAIT_ANY", 0.0) / wave, 3), wait_inst_frac=round(c.get("SQ_WAIT_INST_ANY", 0.0) / wave, 3),
                   active_inst_frac=round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 3))
    if c.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_bank_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    out[k] = row
json.dump(out, open(f"profiles/{tag}_pmc_sq.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"]):
    print(f"{k:40s}", v)
