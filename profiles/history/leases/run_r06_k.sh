#!/bin/bash
# Round 6, lease k: after the 32-bit compare keys are back in the 2^32-bucket instantiation -- r05.so against the tree (with the SNP view, and with VG_NO_SSEC=1), both genomes, same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_k
mkdir -p $OUT
cd $R
one() {
	timeout 900 python3 bench.py --steps 20 --warmup 5 --secondary none --job-reads 0 --no-ingest --cpu-sample 0 --sustain-seconds 0 --no-gather-probe --no-pretouch $2 > $OUT/$1.json 2> $OUT/$1.err
	python3 - $OUT/$1.json $1 <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
o = j.get("other_input_form") or {}
print("%-26s ms/step %.3f kernel %.3f pack %.3f | gate words ms/step %s" % (sys.argv[2], j["ms_per_step"], j["roofline"]["kernel_ms"], j["device_ms_per_step"]["pack"], o.get("ms_per_step")))
PY
}
for g in "--repeats 0.3" ""; do
	tag=default; [ -n "$g" ] && tag=repeats30
	VARGENO_HIP_LIB=$R/variants/r05.so one r05_$tag "$g"
	one tree_$tag "$g"
	VG_NO_SSEC=1 one tree_nossec_$tag "$g"
	VARGENO_HIP_LIB=$R/variants/r05.so one r05_again_$tag "$g"
	one tree_again_$tag "$g"
done
