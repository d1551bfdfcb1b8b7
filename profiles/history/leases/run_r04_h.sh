#!/bin/bash
# Round 4, lease H: the driver's command at full scale (with the secondary legs), the fused-encode A/B at hg38 scale, and the
# profiling recipe (kernel trace + separate --pmc passes) for the default workload, chr22 and the repeat-rich genome.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_h
mkdir -p $OUT
cd $R
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -25 $OUT/bench_default.err | cut -c1-300
ab() {
	local name=$1; shift
	timeout 900 python3 bench.py --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --steps 20 --warmup 5 "$@" > $OUT/$name.json 2> $OUT/$name.err
	python3 - $OUT/$name.json $name <<'PY' | tee -a $OUT/ab_summary.txt
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    d = j["device_ms_per_step"]
    o = j["other_input_form"]
    print("%-12s reads/s %.4g  ms/step %.3f  wave %.3f  pack %.3f  tiers %.3f  frac %.3f  spilled %s | gate words: %.4g  ms/step %.3f wave %.3f pack %.3f" % (sys.argv[2], j["value"], j["ms_per_step"], d["wave"], d["pack"], d["spill_tiers_overlapped"], j["roofline"]["frac"], j.get("reads_per_step_redone_by_deep_list_tier"), o["value"], o["ms_per_step"], o["wave_ms"], o["pack_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
ab base
VARGENO_HIP_LIB=$R/variants/fuse.so ab fuse
ab base2
VARGENO_HIP_LIB=$R/variants/fuse.so ab fuse2
bash profiles/run_prof_r04.sh r04 > $OUT/prof_default.log 2>&1
bash profiles/run_prof_r04.sh r04_chr22 --workload chr22 --steps 40 > $OUT/prof_chr22.log 2>&1
bash profiles/run_prof_r04.sh r04_repeats30 --repeats 0.3 > $OUT/prof_repeats30.log 2>&1
ls $R/gpurun_out/prof_r04* | head -40
df -h /tmp /dev/shm | tail -3
