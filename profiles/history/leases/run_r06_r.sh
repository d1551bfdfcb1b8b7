#!/bin/bash
# Round 6, lease r: the profiling recipe for every workload on the final build (kernel traces + the three --pmc passes each).
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/vg_bench_job /tmp/pytest-of-* 2>/dev/null
cd $R
bash profiles/run_prof_r06.sh r06
bash profiles/run_prof_r06.sh r06_lowq50 --lowq 0.5
bash profiles/run_prof_r06.sh r06_len250 --read-len 250
bash profiles/run_prof_r06.sh r06_chr22 --workload chr22 --steps 40
bash profiles/run_prof_r06.sh r06_chr22_compact --workload chr22 --steps 40 --device-budget 10000000000
bash profiles/run_prof_r06.sh r06_repeats30 --repeats 0.3
rm -rf /tmp/vg_bench/g3100000000_s10000000_c24_r0.3
bash profiles/run_prof_r06.sh r06_hg38f --workload hg38f
rm -rf /dev/shm/vg_bench /tmp/vg_bench/g3100000000_s100000000_c24
ls gpurun_out/ | grep -c traffic
