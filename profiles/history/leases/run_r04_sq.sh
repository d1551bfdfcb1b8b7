#!/bin/bash
# Round 4: what the main-tier kernel's waves do with their cycles (SQ counters, separate --pmc passes of one chr22-scale bench command per group),
# default genome against the repeat-rich one.  Raw output -> gpurun_out/sq_r04/; condensed by the python at the end into sq_<tag>.txt.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
CGRP=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_IFETCH" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum")
for rep in 0 0.3; do
	tag=$( [ $rep = 0 ] && echo def || echo rep30 )
	CMD="python3 $R/bench.py --workload chr22 --repeats $rep --steps 10 --warmup 2 --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0"
	g=0
	for grp in "${CGRP[@]}"; do
		ok=""
		for c in $grp; do grep -qw "$c" $OUT/avail.txt && ok="$ok $c"; done
		if [ -n "$ok" ]; then timeout 300 rocprofv3 --pmc $ok --output-format csv -d $OUT/${tag}_g$g -- $CMD > $OUT/${tag}_g$g.json 2> $OUT/${tag}_g$g.err; fi
		g=$((g+1))
	done
done
python3 - $OUT <<'PY' | tee $OUT/sq_summary.txt
import csv, glob, os, sys, collections
out = sys.argv[1]
for tag in ("def", "rep30"):
    tot = collections.OrderedDict()
    for d in sorted(glob.glob(os.path.join(out, tag + "_g*"))):
        if not os.path.isdir(d): continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc, n = collections.defaultdict(float), collections.defaultdict(int)
            for r in csv.DictReader(open(f)):
                if "vg_wave_kernel<false, 14" not in r.get("Kernel_Name", ""): continue
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
            for k in acc: tot[k] = (acc[k] / max(n[k], 1), n[k])
    print("==", tag, "(per launch of the main-tier kernel, 1 M reads)")
    for k, (v, n) in tot.items(): print("%-40s %18.1f   (%d launches)" % (k, v, n))
PY
rm -rf $OUT/*/*/*agent_info.csv
du -sh $OUT
