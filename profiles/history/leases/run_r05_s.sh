#!/bin/bash
# Round 5, lease s: the round's profiles (profiles/run_prof_r05.sh: kernel trace + three counter passes + the stage clocks per
# workload) on the shipped library -- default, lowq50, chr22, repeats30, hg38f.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for spec in "r05" "r05_lowq50 --lowq 0.5" "r05_chr22 --workload chr22 --steps 40" "r05_repeats30 --repeats 0.3" "r05_hg38f --workload hg38f"; do
	set -- $spec
	tag=$1
	( time bash profiles/run_prof_r05.sh $spec ) > gpurun_out/prof_$tag.log 2>&1
	tail -4 gpurun_out/prof_$tag.log
	grep "traffic_bytes_per_launch" gpurun_out/prof_$tag/traffic_$tag.json
	grep -h "vg_wave_kernel\|vg_pack_kernel" gpurun_out/prof_$tag/summary_$tag.txt | grep calls | head -6
	case $tag in r05_repeats30|r05_hg38f) rm -rf /dev/shm/vg_bench;; esac
	df -B1G --output=used / /dev/shm | tail -2 | tr '\n' ' '; echo
done
