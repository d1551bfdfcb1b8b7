#!/bin/bash
# Round 3, closing lease on the final build: the whole GPU suite + smoke() as the driver runs them, then the profile recipe.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r03_close
rm -rf /tmp/vg_bench /dev/shm/vg_bench
( time python3 -m pytest tests -x -q -m gpu --durations=6 ) > gpurun_out/r03_close/pytest_gpu.log 2>&1
tail -14 gpurun_out/r03_close/pytest_gpu.log
( time python3 -c 'import __graft_entry__ as g; g.smoke()' ) > gpurun_out/r03_close/smoke.log 2>&1
tail -3 gpurun_out/r03_close/smoke.log | head -1
rm -rf /tmp/vg_bench /dev/shm/vg_bench
bash profiles/run_prof_r03.sh r03d
