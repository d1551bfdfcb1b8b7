#!/bin/bash
# Round 4, second closing lease (after the row / stage-B1 work): the driver's command at full scale (with the secondary legs) and the
# profiling recipe (kernel trace + separate --pmc passes) for all five workloads, on the shipped build, most important first.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_final2
mkdir -p $OUT
cd $R
( time timeout 1750 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -12 $OUT/bench_default.err | cut -c1-260
timeout 900 bash profiles/run_prof_r04.sh r04 > $OUT/prof_default.log 2>&1
timeout 1200 bash profiles/run_prof_r04.sh r04_repeats30 --repeats 0.3 > $OUT/prof_repeats30.log 2>&1
timeout 400 bash profiles/run_prof_r04.sh r04_chr22 --workload chr22 --steps 40 > $OUT/prof_chr22.log 2>&1
timeout 600 bash profiles/run_prof_r04.sh r04_lowq50 --lowq 0.5 > $OUT/prof_lowq50.log 2>&1
timeout 900 bash profiles/run_prof_r04.sh r04_hg38f --workload hg38f > $OUT/prof_hg38f.log 2>&1
for t in r04 r04_repeats30 r04_chr22 r04_lowq50 r04_hg38f; do ls $R/gpurun_out/prof_$t/summary_$t.txt $R/gpurun_out/prof_$t/traffic_$t.json; done
