#!/bin/bash
# Development aid: A/B of kernel variants / knobs on the hg38-scale workload within ONE box lease (the index is built once).
#   bash profiles/ab_hg38_r03.sh <tag>    -> gpurun_out/ab_<tag>/*.json + summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ab}
OUT=$R/gpurun_out/ab_$TAG
mkdir -p $OUT
cd $R
ARGS="--cpu-sample 0 --no-gather-probe --no-ingest --steps 20 --warmup 5"
run() {   # name, env assignments...
	local name=$1; shift
	env "$@" VARGENO_VERBOSE=1 python3 bench.py $ARGS > $OUT/$name.json 2> $OUT/$name.err
	python3 - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    d = j["device_ms_per_step"]
    print("%-12s reads/s %.4g  ms/step %.3f  wave %.3f  pack %.3f  tiers %.3f (deep %.3f)  frac %.3f  HBM %.1f GB  views %s  spilled %s" % (sys.argv[2], j["value"], j["ms_per_step"], d["wave"], d["pack"], d["spill_tiers_overlapped"], d["of_which_deep_list_wave_tier"], j["roofline"]["frac"], j["config"]["index_bytes_hbm"] / 1e9, ",".join(j["config"].get("index_views", [])), j.get("reads_per_step_redone_by_deep_list_tier")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
# the shipped build first, with the parity check of the whole 8 M-read batch against the oracle (both builds of the kernel)
ARGS_KEEP=$ARGS; ARGS="--cpu-sample 1000 --cpu-reference no --no-gather-probe --no-ingest --steps 20 --warmup 5"
run base
grep -E "parity" $OUT/base.err | tee -a $OUT/summary.txt
ARGS=$ARGS_KEEP
grep -E "vargeno index|\[vargeno index\]|FASTA|resident" $OUT/base.err | tee -a $OUT/summary.txt
for v in $R/variants/*.so; do
	n=$(basename $v .so)
	case $n in clk*) continue;; esac
	run $n VARGENO_HIP_LIB=$v
done
run base2
if [ -f $R/variants/clk.so ]; then
	VARGENO_HIP_LIB=$R/variants/clk.so python3 bench.py --cpu-sample 0 --no-gather-probe --no-ingest --steps 1 --warmup 0 > $OUT/clk.txt 2> $OUT/clk.err
	grep "dbg" $OUT/clk.err | tail -1 | tee -a $OUT/summary.txt
fi
if [ -z "$AB_NO_TESTS" ]; then ( time python3 -m pytest tests -x -q -m gpu -k "not hg38" ) > $OUT/pytest_all.log 2>&1; fi
grep -E "passed|failed" $OUT/pytest_all.log | tail -2 | tee -a $OUT/summary.txt
