#!/usr/bin/env python3
"""Condense a run_prof_rNN.sh output directory into small text/CSV summaries (stdout + gpurun_out copy)."""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
# which kernel the traffic file is about: the main tier of the read loop ("vg_wave_kernel<false, " ... ", 4>"), or e.g. "vg_wave_kernel_big<"
ksel = sys.argv[3] if len(sys.argv) > 3 else "vg_wave_kernel<false, "


def is_main(name):
    # the main tier's instantiation: four waves per workgroup ("..., 4>" through round 5; "..., 4, false>" / "..., 4, true>" since the SDX template argument of round 6)
    import re
    return ksel in name and re.search(r", 4(, (false|true))?>", name) is not None


out = []
for f in glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")):
    out.append("== rocprofv3 --kernel-trace --stats (%s)" % os.path.relpath(f, src))
    for row in csv.DictReader(open(f)):
        out.append("%-40s calls=%-4s avg_ns=%-12s total_ns=%-12s pct=%s" % (row["Name"].split("(")[0][-40:], row["Calls"], row["AverageNs"], row["TotalDurationNs"], row["Percentage"]))
def per_dispatch(f, want=None):
    """{(kernel, counter): [value per dispatch]} -- a counter without a _sum form comes as one row per instance: added up"""
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if want is not None and not want(r["Kernel_Name"]):
            continue
        k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"], r["Dispatch_Id"])
        acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
    agg = collections.defaultdict(list)
    for (kn, cn, _), v in acc.items():
        agg[(kn, cn)].append(v)
    return agg


for grp in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_ea", "pmc_sq"):
    for f in glob.glob(os.path.join(src, grp, "*", "*_counter_collection.csv")):
        agg = per_dispatch(f)
        out.append("== rocprofv3 --pmc (%s): mean per dispatch" % grp)
        for (k, c), v in sorted(agg.items()):
            if "vg_" in k:
                out.append("%-40s %-22s n=%-3d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
for name in ("bench_default.json", "kt.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        out.append("== %s" % name)
        out.append(open(p).read().strip())
# traffic of the dominant kernel for bench.py's roofline.traffic: (FETCH_SIZE + WRITE_SIZE) KB per launch, separate --pmc passes
means = {}
for grp in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_ea"):
    for f in glob.glob(os.path.join(src, grp, "*", "*_counter_collection.csv")):
        for (kn, c), v in per_dispatch(f, is_main).items():
            means[c] = sum(v) / len(v)
kt_avg, kname = None, None
for f in glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        if is_main(row["Name"]):
            kt_avg = float(row["AverageNs"])
            kname = row["Name"].split("(")[0].replace("void ", "").replace("vg::", "").strip()
if "WRITE_SIZE" in means and ("TCC_EA0_RDREQ_DRAM_32B" in means or "TCC_MISS_sum" in means):
    wl, build_id = None, None
    try:
        cfg = json.loads(open(os.path.join(src, "kt.json")).read().strip().splitlines()[-1])["config"]
        wl = {"genome": cfg["genome_bp"], "snps": cfg["snps_requested"], "reads": cfg["reads_per_step_per_gpu"], "lowq": cfg.get("lowq", 0.08), "repeats": cfg.get("repeats", 0.0), "gate_words": cfg.get("gate_words", True),
              "read_len": cfg.get("read_len", 150), "softmask": cfg.get("softmask", 0.0), "device_budget": cfg.get("device_budget")}
        build_id = cfg.get("lib_build_id")
    except Exception:
        pass
    # What an L2 miss moves was measured in round 3 (profiles/line_probe_r03_counters.txt, fetch_gran_probe_r03_counters.txt): ONE
    # 128-byte request to the fabric = four 32-byte DRAM requests, for every load flavour and allocation type.  FETCH_SIZE tallies
    # such a request at 64 bytes (its formula prices 128-byte requests through TCC_BUBBLE, which stays 0 on gfx950), so the read
    # traffic is taken from TCC_EA0_RDREQ_DRAM_32B x 32 (else TCC_MISS_sum x 128); WRITE_SIZE counts stores and atomics exactly.
    if "TCC_EA0_RDREQ_DRAM_32B" in means:
        read_bytes, formula = means["TCC_EA0_RDREQ_DRAM_32B"] * 32, "TCC_EA0_RDREQ_DRAM_32B x 32 B + WRITE_SIZE"
    else:
        read_bytes, formula = means["TCC_MISS_sum"] * 128, "TCC_MISS_sum x 128 B + WRITE_SIZE"
    tj = {"workload": wl, "build_id": build_id, "kernel": kname or "vg_wave_kernel (main tier)", "FETCH_SIZE_KB": means.get("FETCH_SIZE"), "WRITE_SIZE_KB": means["WRITE_SIZE"],
          "traffic_bytes_per_launch": int(read_bytes + means["WRITE_SIZE"] * 1024), "traffic_formula": formula,
          "TCC_EA0_RDREQ_128B": means.get("TCC_EA0_RDREQ_128B"), "TCC_EA0_RDREQ_64B": means.get("TCC_EA0_RDREQ_64B"), "TCC_EA0_RDREQ_DRAM_32B": means.get("TCC_EA0_RDREQ_DRAM_32B"),
          "TCC_MISS_sum": means.get("TCC_MISS_sum"), "TCC_HIT_sum": means.get("TCC_HIT_sum"), "TCP_TCC_READ_REQ_sum": means.get("TCP_TCC_READ_REQ_sum"),
          "kernel_trace_avg_ns": kt_avg,
          "source": "profiles/run_prof_r06.sh %s -> profiles/rocprof_summary_%s.txt; what one L2 miss moves: profiles/line_probe_r03.*" % (tag, tag)}
    open(os.path.join(src, "traffic_%s.json" % tag), "w").write(json.dumps(tj, indent=1) + "\n")
    out.append("== traffic_%s.json" % tag)
    out.append(json.dumps(tj))
txt = "\n".join(out)
print(txt)
open(os.path.join(src, "summary_%s.txt" % tag), "w").write(txt + "\n")
