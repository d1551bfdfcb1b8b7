#!/bin/bash
# Round 4, lease J: kernel timeline of the default workload (rocprofv3 --kernel-trace, trace kept): when do the tiers of batch k run?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 12 --warmup 2 --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 > $OUT/kt.json 2> $OUT/kt.err
python3 - $OUT <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
f = glob.glob(os.path.join(out, "kt", "*", "*_kernel_trace.csv"))[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "vg_wave_kernel" in n or "vg_pack" in n or "vg_lane" in n or "accumulate" in n or "fold" in n or "clamp" in n:
        short = n.split("(")[0].replace("void ", "").replace("vg::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
t0 = rows[0][0]
with open(os.path.join(out, "timeline.txt"), "w") as g:
    for a, b, n, q in rows:
        g.write("%10.3f %10.3f %8.3f  q%-3s %s\n" % ((a - t0) / 1e6, (b - t0) / 1e6, (b - a) / 1e6, q, n))
print(open(os.path.join(out, "timeline.txt")).read()[-6000:])
PY
rm -rf $OUT/kt
