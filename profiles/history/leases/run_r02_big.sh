#!/bin/bash
# (development) default-config A/B of the shipped library against variants/prev.so, the GPU tests without the hg38-scale ones,
# then the configs[4]-scale single-replica run
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash profiles/ab_hg38_r02.sh a11
rm -rf /tmp/vg_bench
bash profiles/run_hg38f_r02.sh
