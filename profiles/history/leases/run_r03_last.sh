#!/bin/bash
# Round 3, closing lease: the whole GPU suite + smoke() exactly as the driver runs them (no VG_BENCH_DIR: the hg38 index goes to
# /tmp, the hg38 + 100 M SNPs one to /dev/shm), then configs[4]'s index on one replica: bench + rocprofv3 of vg_wave_kernel_big.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_last
mkdir -p $OUT
cd $R
rm -rf /tmp/vg_bench /dev/shm/vg_bench
( time python3 -m pytest tests -x -q -m gpu ) > $OUT/pytest_gpu.log 2>&1
tail -6 $OUT/pytest_gpu.log
( time python3 -c 'import __graft_entry__ as g; g.smoke()' ) > $OUT/smoke.log 2>&1
tail -3 $OUT/smoke.log
rm -rf /tmp/vg_bench
SKIP_TEST=1 bash profiles/run_hg38f_r03.sh
