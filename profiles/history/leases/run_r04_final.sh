#!/bin/bash
# Round 4, closing lease: the driver's command at full scale (with the secondary legs) and the profiling recipe (kernel trace +
# separate --pmc passes) for the default workload, chr22 and the repeat-rich genome, all on the shipped build.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_final
mkdir -p $OUT
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_fastq.py tests/test_gpu_cli.py -k "not hg38" -x -q -m gpu ) > $OUT/pytest_fastq_cli.log 2>&1; tail -3 $OUT/pytest_fastq_cli.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -22 $OUT/bench_default.err | cut -c1-300
bash profiles/run_prof_r04.sh r04 > $OUT/prof_default.log 2>&1
bash profiles/run_prof_r04.sh r04_chr22 --workload chr22 --steps 40 > $OUT/prof_chr22.log 2>&1
bash profiles/run_prof_r04.sh r04_repeats30 --repeats 0.3 > $OUT/prof_repeats30.log 2>&1
for t in r04 r04_chr22 r04_repeats30; do grep -c . $R/gpurun_out/prof_$t/summary_$t.txt; done
