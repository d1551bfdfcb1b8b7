#!/bin/bash
# Round 5: the profile recipe for 250 bp reads (traffic file of the len250 leg).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
( time bash profiles/run_prof_r05.sh r05_len250 --read-len 250 ) > gpurun_out/prof_r05_len250.log 2>&1
tail -3 gpurun_out/prof_r05_len250.log
grep "traffic_bytes_per_launch\|kernel_trace_avg_ns" gpurun_out/prof_r05_len250/traffic_r05_len250.json
grep -h "vg_wave_kernel\|vg_pack_kernel\|vg_lane" gpurun_out/prof_r05_len250/summary_r05_len250.txt | grep calls | head -8
