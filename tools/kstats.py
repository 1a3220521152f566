#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 kernel_stats.csv found under a directory:  python3 tools/kstats.py <dir> [rows]"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in list(csv.DictReader(open(f)))[:n]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"\(.*", "", re.sub(r"^void ", "", name))
    print("%-56s calls %5s  avg %8.1f us  min %8.1f  max %8.1f  %s%%" % (name[:56], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                                                                       float(r["MaxNs"]) / 1e3, r["Percentage"]))
