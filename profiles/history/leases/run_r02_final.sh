#!/bin/bash
# Round-2 closing validation on one box lease: the whole GPU test-suite (what the driver runs), smoke(), the default bench, the
# chr22-scale bench, and `vargeno geno` end to end at hg38 scale.   -> gpurun_out/final_r02/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final_r02
mkdir -p $OUT
cd $R
( time python3 -m pytest tests -x -q -m gpu ) > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
python3 -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -4 $OUT/bench_default.err
( time python3 bench.py --workload chr22 --steps 20 --warmup 5 ) > $OUT/bench_chr22.json 2> $OUT/bench_chr22.err
bash profiles/run_cli_hg38_r02.sh
cp $R/gpurun_out/cli_hg38/*.log $OUT/ 2>/dev/null
