#!/bin/bash
# Development aid: one bench workload under different environment knobs / library variants within ONE box lease (the index is
# built once).   BENCH_ARGS="--repeats 0.3" bash profiles/ab_env_r03.sh <tag> NAME:VAR=value[,VAR=value...] ...
#   -> gpurun_out/ab_<tag>/*.json + summary.txt        (VARGENO_HIP_LIB=variants/x.so selects a library variant; paths relative to the repo)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-env}; shift
OUT=$R/gpurun_out/ab_$TAG
mkdir -p $OUT
cd $R
ARGS="--cpu-sample 0 --no-gather-probe --no-ingest --steps 20 --warmup 5 $BENCH_ARGS"
run() {   # name, env assignments...
	local name=$1; shift
	env "$@" VARGENO_VERBOSE=1 python3 bench.py $ARGS > $OUT/$name.json 2> $OUT/$name.err
	python3 - $OUT/$name.json $name <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    d = j["device_ms_per_step"]
    print("%-14s reads/s %.4g  ms/step %.3f  wave %.3f  pack %.3f  tiers %.3f (deep lists %.3f)  frac %.3f  redone %s  lane tier %s" % (sys.argv[2], j["value"], j["ms_per_step"], d["wave"], d["pack"], d["spill_tiers_overlapped"], d["of_which_deep_list_wave_tier"], j["roofline"]["frac"],
          j.get("reads_per_step_redone_by_deep_list_tier"), j.get("reads_per_step_sent_on_to_lane_tier")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
[ -n "$AB_CLEAN_TMP" ] && rm -rf /tmp/vg_bench
run base
grep -E "vargeno index:|resident|parity" $OUT/base.err | tee -a $OUT/summary.txt
for spec in "$@"; do
	name=${spec%%:*}
	run $name $(echo ${spec#*:} | tr ',' ' ' | sed "s|VARGENO_HIP_LIB=|VARGENO_HIP_LIB=$R/|")
done
run base2
