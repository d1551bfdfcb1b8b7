#!/bin/bash
# Round 6, lease l: large blocks scanned instead of queried -- parity (fixtures with large blocks, every layout), then the tree against itself without the scan (variants/nolscan.so) and round 5's library.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_l
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/tests.txt 2>&1
tail -4 $OUT/tests.txt
one() {
	timeout 900 python3 bench.py --steps 20 --warmup 5 --secondary none --job-reads 0 --no-ingest --cpu-sample 0 --sustain-seconds 0 --no-gather-probe --no-pretouch $2 > $OUT/$1.json 2> $OUT/$1.err
	python3 - $OUT/$1.json $1 <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
o = j.get("other_input_form") or {}
print("%-26s ms/step %.3f kernel %.3f pack %.3f | gate words ms/step %s" % (sys.argv[2], j["ms_per_step"], j["roofline"]["kernel_ms"], j["device_ms_per_step"]["pack"], o.get("ms_per_step")))
PY
}
for g in "--repeats 0.3" "" "--lowq 0.5"; do
	tag=default; [ "$g" = "--repeats 0.3" ] && tag=repeats30; [ "$g" = "--lowq 0.5" ] && tag=lowq50
	VARGENO_HIP_LIB=$R/variants/r05.so one r05_$tag "$g"
	VARGENO_HIP_LIB=$R/variants/nolscan.so one nolscan_$tag "$g"
	one tree_$tag "$g"
	VARGENO_HIP_LIB=$R/variants/nolscan.so one nolscan_again_$tag "$g"
	one tree_again_$tag "$g"
done
