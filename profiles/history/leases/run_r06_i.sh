#!/bin/bash
# Round 6, lease i: where did 2.6 % of the default kernel (7 % on the repeat-rich genome) go between round 5's library and this tree with VG_NO_SSEC=1?
# Same box: r05.so, the tree, the tree without the coarse jump table's code, without the SNP view's code, without either; VG_NO_SSEC=1 everywhere the view exists (so that code, not the view, is compared).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_i
mkdir -p $OUT
cd $R
one() {
	timeout 900 python3 bench.py --steps 20 --warmup 5 --secondary none --job-reads 0 --no-ingest --cpu-sample 0 --sustain-seconds 0 --no-gather-probe --no-pretouch $2 > $OUT/$1.json 2> $OUT/$1.err
	python3 - $OUT/$1.json $1 <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
d = json.load(open(j["detail"]))
print("%-26s ms/step %.3f kernel %.3f pack %.3f tail %.3f deep %.3f" % (sys.argv[2], j["ms_per_step"], j["roofline"]["kernel_ms"], j["device_ms_per_step"]["pack"], d["device_ms_per_step"]["spill_tiers_overlapped"], d["device_ms_per_step"]["of_which_deep_list_wave_tier"]))
PY
}
export VG_NO_SSEC=1
for g in "" "--repeats 0.3"; do
	tag=default; [ -n "$g" ] && tag=repeats30
	VARGENO_HIP_LIB=$R/variants/r05.so one r05_$tag "$g"
	one tree_$tag "$g"
	VG_W2_FULL_GRID=1 one tree_fullgrid_$tag "$g"
	VG_NO_LATE_STORE=1 one tree_nolate_$tag "$g"
	VARGENO_HIP_LIB=$R/variants/nocoarse.so one nocoarse_$tag "$g"
	VARGENO_HIP_LIB=$R/variants/nossecc.so one nossecc_$tag "$g"
	VARGENO_HIP_LIB=$R/variants/neither.so one neither_$tag "$g"
done
