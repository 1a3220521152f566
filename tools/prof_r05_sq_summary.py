#!/usr/bin/env python3
"""Condense the two SQ counter passes of tools/prof_r05_sq.sh into <outdir>/<tag>_pmc_sq_<workload>.json: per kernel (every kernel with
>= 0.5 % of the workload's GPU time), averages per launch and the busy fractions of the matrix pipes, the LDS array, the VALU and the
instruction issue.

  python tools/prof_r05_sq_summary.py <tag> <workload> <outdir> <passA counter_collection.csv> [<passB counter_collection.csv>]

Units (MI355X_MICROARCH.md, rocprofv3 PMC notes): SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's 1024 SIMDs
(16 per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_LDS_IDX_ACTIVE counts LDS-array cycles summed over the 256 CUs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (chip cycles = / 8)."""
import collections
import csv
import json
import os
import re
import sys

tag, workload, outdir, path_a = sys.argv[1:5]
path_b = sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] else None


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def read(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    ids = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    if not path or not os.path.exists(path):
        return acc, ids, dur
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in ids[k]:
            ids[k].add(r["Dispatch_Id"])
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return acc, ids, dur


# measured in-kernel clocks (tools/lab/clock_probe.py -> profiles/r06_clock.json), by kernel name
CLOCKS = {}
_cj = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_clock.json")
if os.path.exists(_cj):
    for name, v in json.load(open(_cj)).get("in_model_mhz", {}).items():
        CLOCKS[name] = v
acc, ids, dur = read(path_a)
accb, idsb, _ = read(path_b)
total = sum(dur.values()) or 1.0
out = {}
for k, c in acc.items():
    n = len(ids[k])
    if dur[k] / total < 0.005 or n == 0:
        continue
    us = dur[k] / n / 1e3
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / n / 8.0                     # chip cycles per launch
    mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n
    wave = c.get("SQ_WAVE_CYCLES", 0.0)
    # GRBM_GUI_ACTIVE / 8 / wall is NOT a clock for launches this short (MI355X_MICROARCH.md, DVFS: "reads high on dispatches shorter than
    # about 0.3 ms" -- round 5 printed 2.3-15.8 GHz from it): it is kept as what it is, the counter per microsecond, and nothing is derived
    # from it.  Busy fractions are given against the launch's wall time at the nominal 2.4 GHz (the clock the 2.5 PFLOP/s peak is quoted
    # on: the roofline fraction of a pure-MFMA kernel) and, where tools/lab/clock_probe.py measured the kernel's own clock
    # (profiles/r06_clock.json: s_memtime / s_memrealtime inside the kernel), against the cycles the launch really had.
    mhz = CLOCKS.get(k.split("<")[0])
    row = dict(launches_profiled=n, share_of_gpu_time=round(dur[k] / total, 4), avg_us_under_pmc=round(us, 1),
               grbm_gui_active_per_us=round(cyc / us) if us else None,
               in_kernel_clock_mhz=mhz,
               mfma_busy_frac=round(mfma / (1024 * us * mhz), 4) if (us and mhz) else None,
               mfma_busy_frac_at_2p4ghz=round(mfma / (1024 * us * 1e-6 * 2.4e9), 4) if us else None,
               lds_busy_frac_at_2p4ghz=round(c.get("SQ_LDS_IDX_ACTIVE", 0.0) / n / (256 * us * 1e-6 * 2.4e9), 4) if us else None,
               lds_busy_frac=round(c.get("SQ_LDS_IDX_ACTIVE", 0.0) / n / (256 * us * mhz), 4) if (us and mhz) else None)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_bank_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    if wave:
        row.update(wait_any_frac=round(c.get("SQ_WAIT_ANY", 0.0) / wave, 3), wait_inst_frac=round(c.get("SQ_WAIT_INST_ANY", 0.0) / wave, 3),
                   active_inst_frac=round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 3))
    cb = accb.get(k)
    if cb and cb.get("SQ_WAVE_CYCLES"):
        wb = cb["SQ_WAVE_CYCLES"]
        row.update(valu_inst_frac=round(cb.get("SQ_ACTIVE_INST_VALU", 0.0) / wb, 3), lds_inst_frac=round(cb.get("SQ_ACTIVE_INST_LDS", 0.0) / wb, 3),
                   vmem_inst_frac=round(cb.get("SQ_ACTIVE_INST_VMEM", 0.0) / wb, 3), salu_inst_frac=round(cb.get("SQ_ACTIVE_INST_SCA", 0.0) / wb, 3))
    out[k] = row
os.makedirs(outdir, exist_ok=True)
json.dump(out, open(os.path.join(outdir, f"{tag}_pmc_sq_{workload}.json"), "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["share_of_gpu_time"]):
    print(f"{k[:44]:44s} {100 * v['share_of_gpu_time']:5.1f}%  {v['avg_us_under_pmc']:8.1f} us  mfma@2.4GHz {v['mfma_busy_frac_at_2p4ghz']}  mfma@own clock {v['mfma_busy_frac']}  lds {v['lds_busy_frac_at_2p4ghz']}  "
          f"valu {v.get('valu_inst_frac')}  wait_any {v.get('wait_any_frac')}  wait_inst {v.get('wait_inst_frac')}")
