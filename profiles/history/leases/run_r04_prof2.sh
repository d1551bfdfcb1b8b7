#!/bin/bash
# Round 4: the profiling recipe for the two remaining secondary workloads (stress profile; configs[4]'s index) on the shipped build.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash profiles/run_prof_r04.sh r04_lowq50 --lowq 0.5 > $R/gpurun_out/prof2_lowq50.log 2>&1
tail -2 $R/gpurun_out/prof2_lowq50.log
bash profiles/run_prof_r04.sh r04_hg38f --workload hg38f > $R/gpurun_out/prof2_hg38f.log 2>&1
tail -2 $R/gpurun_out/prof2_hg38f.log
for t in r04_lowq50 r04_hg38f; do ls -la $R/gpurun_out/prof_$t/ | head -20; done
