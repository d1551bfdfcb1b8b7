#!/usr/bin/env python3
"""Condense a run_prof_rNN.sh output directory into small text/CSV summaries (stdout + gpurun_out copy)."""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
out = []
for f in glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")):
    out.append("== rocprofv3 --kernel-trace --stats (%s)" % os.path.relpath(f, src))
    for row in csv.DictReader(open(f)):
        out.append("%-40s calls=%-4s avg_ns=%-12s total_ns=%-12s pct=%s" % (row["Name"].split("(")[0][-40:], row["Calls"], row["AverageNs"], row["TotalDurationNs"], row["Percentage"]))
for grp in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq"):
    for f in glob.glob(os.path.join(src, grp, "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        out.append("== rocprofv3 --pmc (%s): mean per dispatch" % grp)
        for (k, c), v in sorted(agg.items()):
            if "vg_" in k:
                out.append("%-40s %-22s n=%-3d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
for name in ("bench_default.json", "kt.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        out.append("== %s" % name)
        out.append(open(p).read().strip())
txt = "\n".join(out)
print(txt)
open(os.path.join(src, "summary_%s.txt" % tag), "w").write(txt + "\n")
