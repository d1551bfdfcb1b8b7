#!/bin/bash
# Round 6, lease g: the shipped library against round 5's (variants/r05.so, built from commit 2f5b27e) and against itself without the SNP view, same box,
# default workload (timing only: --cpu-sample 0), then the repeat-rich genome.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_g
mkdir -p $OUT
cd $R
one() {   # tag, extra bench args; env from the caller
	timeout 900 python3 bench.py --steps 20 --warmup 5 --secondary none --job-reads 0 --no-ingest --cpu-sample 0 --sustain-seconds 0 --no-gather-probe --no-pretouch $2 > $OUT/$1.json 2> $OUT/$1.err
	python3 - $OUT/$1.json $1 <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
o = j.get("other_input_form") or {}
print("%-22s value %.4g ms/step %.3f kernel %.3f pack %.3f | gate words: ms/step %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["device_ms_per_step"]["pack"], o.get("ms_per_step")))
PY
}
for rep in 1 2; do
	VARGENO_HIP_LIB=$R/variants/r05.so one r05_default_$rep ""
	one r06_default_$rep ""
	VG_NO_SSEC=1 one r06_nossec_default_$rep ""
done
VARGENO_HIP_LIB=$R/variants/r05.so one r05_repeats30 "--repeats 0.3"
one r06_repeats30 "--repeats 0.3"
