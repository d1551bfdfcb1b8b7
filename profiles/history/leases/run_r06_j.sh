#!/bin/bash
# Round 6, lease j: bisect the main kernel's slowdown over the round's library commits (variants/c_<commit>.so), repeat-rich genome then default.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_j
mkdir -p $OUT
cd $R
one() {
	timeout 900 python3 bench.py --steps 20 --warmup 5 --secondary none --job-reads 0 --no-ingest --cpu-sample 0 --sustain-seconds 0 --no-gather-probe --no-pretouch $2 > $OUT/$1.json 2> $OUT/$1.err
	python3 - $OUT/$1.json $1 <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%-26s ms/step %.3f kernel %.3f pack %.3f" % (sys.argv[2], j["ms_per_step"], j["roofline"]["kernel_ms"], j["device_ms_per_step"]["pack"]))
PY
}
export VG_NO_SSEC=1
for g in "--repeats 0.3" ""; do
	tag=default; [ -n "$g" ] && tag=repeats30
	for lib in r05 c_e2eaf85 c_af633e9 c_d299ddd; do VARGENO_HIP_LIB=$R/variants/$lib.so one ${lib}_$tag "$g"; done
	one tree_$tag "$g"
	VARGENO_HIP_LIB=$R/variants/r05.so one r05_again_$tag "$g"
done
