#!/bin/bash
# Round 3, secondary numbers on the shipped build: the repeat-rich genome (parity on the 8 M-read batch), the 50 % low-quality
# stress profile of SURVEY.md §8d at hg38 scale, the chr22-scale workload (BASELINE.json configs[1]).  -> gpurun_out/r03_extras/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_extras
mkdir -p $OUT
cd $R
export VARGENO_VERBOSE=1
rm -rf /tmp/vg_bench
( time python3 bench.py --repeats 0.3 --cpu-sample 1000000 --cpu-reference no --no-ingest --no-gather-probe --steps 10 --warmup 2 ) > $OUT/bench_hg38_repeats30.json 2> $OUT/bench_hg38_repeats30.err
grep -E "parity|resident" $OUT/bench_hg38_repeats30.err
rm -rf /tmp/vg_bench
( time python3 bench.py --lowq 0.5 --cpu-sample 200000 --cpu-reference no --no-ingest --no-gather-probe --steps 10 --warmup 2 ) > $OUT/bench_hg38_lowq50.json 2> $OUT/bench_hg38_lowq50.err
grep -E "parity|resident" $OUT/bench_hg38_lowq50.err
( time python3 bench.py --workload chr22 --steps 20 --warmup 5 ) > $OUT/bench_chr22.json 2> $OUT/bench_chr22.err
grep -E "parity|resident|reference" $OUT/bench_chr22.err
for f in $OUT/*.json; do python3 - $f <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); d = j["device_ms_per_step"]
    print("%-30s reads/s %.4g  ms/step %.3f  wave %.3f  pack %.3f  tiers %.3f  frac %.3f  redone %s  lane tier %s  cpu %s" % (sys.argv[1].split("/")[-1], j["value"], j["ms_per_step"], d["wave"], d["pack"], d["spill_tiers_overlapped"], j["roofline"]["frac"],
          j.get("reads_per_step_redone_by_deep_list_tier"), j.get("reads_per_step_sent_on_to_lane_tier"), j["cpu_baseline"] and (j["cpu_baseline"]["kind"], round(j["cpu_baseline"]["value"]))))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
