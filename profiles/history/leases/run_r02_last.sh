#!/bin/bash
# The round's last GPU run, at HEAD: the whole GPU test-suite (log kept), smoke(), `vargeno geno` end to end at hg38 scale.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/last_r02
mkdir -p $OUT
cd $R
( time python3 -m pytest tests -x -q -m gpu ) > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
python3 -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash profiles/run_cli_hg38_r02.sh > $OUT/cli.log 2>&1
cp $R/gpurun_out/cli_hg38/*.log $OUT/ 2>/dev/null
grep -h "reads:" $OUT/geno_host*.log
tail -2 $OUT/cli.log
