#!/bin/bash
# Can a random gather be made to move less than a 128-byte line?  tools/fetch_gran_probe (allocation x load flavour) plain and
# under --pmc passes, one allocation mode per run (the kernels are the same, the allocation differs).  -> gpurun_out/fetch_gran/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fetch_gran
mkdir -p $OUT
P=$R/vargeno_amd/csrc/tools/fetch_gran_probe
cd /tmp && export TMPDIR=/tmp
timeout 300 $P 8 > $OUT/probe.jsonl 2> $OUT/probe.err
cat $OUT/probe.jsonl
for mode in 0 1 2; do
	timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum --output-format csv -d $OUT/a$mode -- $P 8 $mode > $OUT/a$mode.out 2> $OUT/a$mode.err
	timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B TCC_EA0_RDREQ_DRAM_32B TCC_EA0_RD_UNCACHED_32B_sum --output-format csv -d $OUT/b$mode -- $P 8 $mode > $OUT/b$mode.out 2> $OUT/b$mode.err
done
python3 - $OUT <<'PY' | tee $OUT/counters.txt
import csv, glob, sys, collections
out = sys.argv[1]
names = {0: "hipMalloc", 1: "finegrained", 2: "uncached"}
for mode in (0, 1, 2):
    acc = {}
    for f in sorted(glob.glob(out + "/[ab]%d/*/*_counter_collection.csv" % mode)):
        for r in csv.DictReader(open(f)):
            if "fgp<" not in r["Kernel_Name"]: continue
            k = (r["Kernel_Name"].split("(")[0][-14:], r["Counter_Name"])
            d = int(r["Dispatch_Id"])
            if k not in acc or d > acc[k][0]: acc[k] = (d, 0.0)
            if d == acc[k][0]: acc[k] = (d, acc[k][1] + float(r["Counter_Value"]))
    print("== allocation: %s   (per kernel: the last, timed launch; 33 554 432 x ... loads)" % names[mode])
    kn = []
    for (k, c) in acc:
        if k not in kn: kn.append(k)
    for k in kn:
        print("  " + k + "  " + "  ".join("%s=%.0f" % (c.replace("TCC_EA0_", "").replace("_sum", ""), v[1]) for (k2, c), v in acc.items() if k2 == k))
PY
rm -rf $OUT/[ab]*/*/*agent_info.csv
