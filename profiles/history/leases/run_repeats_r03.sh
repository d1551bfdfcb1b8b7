#!/bin/bash
# Repeat-rich stress genome (bench.py --repeats F: that fraction of the genome in planted families of near-identical copies,
# 50 microsatellites per Mbp) at hg38 scale: parity on the 8 M-read batch, reads/s, spilled reads, tail vs main.
#   bash profiles/run_repeats_r03.sh [F=0.3] [more bench args]      -> gpurun_out/repeats_r03/
R=${GRAFT_REPO_ROOT:-$(pwd)}
F=${1:-0.3}; shift
OUT=$R/gpurun_out/repeats_r03
mkdir -p $OUT
cd $R
export VARGENO_VERBOSE=1 VG_VERBOSE=1
rm -rf /tmp/vg_bench   # (an earlier run's 48 GB index: /tmp holds one of them)
A="--repeats $F --cpu-reference no --no-ingest --no-gather-probe --steps 10 --warmup 2"
( time timeout 1500 python3 bench.py $A --cpu-sample 1000000 "$@" ) > $OUT/bench_F$F.json 2> $OUT/bench_F$F.err
tail -6 $OUT/bench_F$F.err
for v in $R/variants/*.so; do
	n=$(basename $v .so)
	VARGENO_HIP_LIB=$v timeout 900 python3 bench.py $A --cpu-sample 0 --steps $([ $n = clk ] && echo 1 || echo 10) --warmup $([ $n = clk ] && echo 0 || echo 2) > $OUT/v_${n}_F$F.json 2> $OUT/v_${n}_F$F.err
done
for f in $OUT/*_F$F.json; do python3 - $f <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); d = j["device_ms_per_step"]
    print("%-36s reads/s %.4g  ms/step %.3f  wave %.3f  pack %.3f  tiers %.3f (deep lists %.3f)  redone %s  lane tier %s  ctx/read %.2f" % (sys.argv[1].split("/")[-1], j["value"], j["ms_per_step"], d["wave"], d["pack"], d["spill_tiers_overlapped"], d["of_which_deep_list_wave_tier"],
          j["reads_per_step_redone_by_deep_list_tier"], j["reads_per_step_sent_on_to_lane_tier"], j["events_per_read"]["ctx"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
grep -h "CLK\|dbg" $OUT/v_clk_F$F.json $OUT/v_clk_F$F.err 2>/dev/null | tail -4
