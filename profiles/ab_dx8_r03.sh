#!/bin/bash
# EXPERIMENT: -DVG_DX8 (variants/dx8.so) against the shipped build on the default workload, parity checked for both.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ab_dx8
mkdir -p $OUT
cd $R
rm -rf /tmp/vg_bench
A="--cpu-reference no --no-gather-probe --no-ingest --steps 20 --warmup 5"
show() { python3 - $1 $2 <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); d = j["device_ms_per_step"]
    print("%-10s reads/s %.4g  ms/step %.3f  wave %.3f  pack %.3f  frac %.3f  redone %s" % (sys.argv[2], j["value"], j["ms_per_step"], d["wave"], d["pack"], j["roofline"]["frac"], j.get("reads_per_step_redone_by_deep_list_tier")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
python3 bench.py $A --cpu-sample 1000 > $OUT/base.json 2> $OUT/base.err; show $OUT/base.json base; grep parity $OUT/base.err | tee -a $OUT/summary.txt
VARGENO_HIP_LIB=$R/variants/dx8.so python3 bench.py $A --cpu-sample 1000 > $OUT/dx8.json 2> $OUT/dx8.err; show $OUT/dx8.json dx8; grep -E "parity|Error|assert" $OUT/dx8.err | tee -a $OUT/summary.txt
python3 bench.py $A --cpu-sample 0 > $OUT/base2.json 2> $OUT/base2.err; show $OUT/base2.json base2
VARGENO_HIP_LIB=$R/variants/dx8.so python3 bench.py $A --cpu-sample 0 > $OUT/dx8b.json 2> $OUT/dx8b.err; show $OUT/dx8b.json dx8b
( VARGENO_HIP_LIB=$R/variants/dx8.so python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu ) > $OUT/pytest_dx8.log 2>&1; tail -3 $OUT/pytest_dx8.log | tee -a $OUT/summary.txt
cd /tmp && export TMPDIR=/tmp
for v in base dx8; do
	[ $v = dx8 ] && export VARGENO_HIP_LIB=$R/variants/dx8.so
	rocprofv3 --pmc TCC_EA0_RDREQ_128B TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc_$v -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-gather-probe --no-ingest > $OUT/pmc_$v.json 2> $OUT/pmc_$v.err
done
python3 - $OUT <<'PY' | tee -a $OUT/summary.txt
import csv, glob, sys, collections
for v in ("base", "dx8"):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for f in glob.glob(sys.argv[1] + "/pmc_%s/*/*_counter_collection.csv" % v):
        for r in csv.DictReader(open(f)):
            if "vg_wave_kernel<false, 14" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    print(v, {k: round(acc[k] / max(1, len(n[k])) / 8e6, 2) for k in acc}, "per read")
PY
rm -rf $OUT/pmc_*/*/*agent_info.csv
