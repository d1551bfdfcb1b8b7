#!/bin/bash
# How many bytes does one L2 miss of a random 8-byte gather move on gfx950?  tools/line_probe under separate --pmc passes
# (program directly after `--`).  Run on the MI355X box from the repo root:  bash profiles/run_line_probe_r03.sh
#   -> gpurun_out/line_probe/{probe.jsonl, counters.txt}   (copied into profiles/line_probe_r03.*)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/line_probe
mkdir -p $OUT
P=$R/vargeno_amd/csrc/tools/line_probe
cd /tmp && export TMPDIR=/tmp
$P 16 > $OUT/probe.jsonl 2> $OUT/probe.err
cat $OUT/probe.jsonl
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B" \
           "TCC_EA0_RDREQ_DRAM_32B TCC_READ_SECTORS" "TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE"; do
	i=$((i + 1))
	timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- $P 16 > $OUT/p$i.out 2> $OUT/p$i.err || echo "pass $i ($grp) failed: $(tail -2 $OUT/p$i.err)"
done
python3 - $OUT <<'PY' | tee $OUT/counters.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.OrderedDict()
for f in sorted(glob.glob(out + "/p*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-28:], int(r["Dispatch_Id"]), r["Counter_Name"])
        acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
# the timed launches: for every kernel the LAST dispatch of each counter pass
last = {}
for (kn, did, cn), v in acc.items():
    if "<5>" in kn: kn = "%s launch #%d (seq_a, seq_b = +64, seq_a again)" % (kn, did)     # the three launches of the small set, one by one
    if (kn, cn) not in last or did >= last[(kn, cn)][0]:
        last[(kn, cn)] = (did, v)
names = []
for (kn, cn) in last:
    if kn not in names: names.append(kn)
for kn in names:
    print(kn)
    for (k2, cn), (did, v) in last.items():
        if k2 == kn: print("    %-28s dispatch %-4d %16.0f" % (cn, did, v))
PY
rm -rf $OUT/p*/*/*agent_info.csv
