#!/bin/bash
# EXPERIMENT: -DVG_CANON (variants/canon.so: merged view keyed by canonical k-mers, a reverse-strand read's one useful pass runs on
# the look-ups of its forward pass) against the shipped build on the default workload, parity checked for both.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ab_canon
mkdir -p $OUT
cd $R
rm -rf /tmp/vg_bench
A="--cpu-reference no --no-gather-probe --no-ingest --steps 20 --warmup 5"
show() { python3 - $1 $2 <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); d = j["device_ms_per_step"]
    print("%-10s reads/s %.4g  ms/step %.3f  wave %.3f  pack %.3f  frac %.3f  redone %s" % (sys.argv[2], j["value"], j["ms_per_step"], d["wave"], d["pack"], j["roofline"]["frac"], j.get("reads_per_step_redone_by_deep_list_tier")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
python3 bench.py $A --cpu-sample 0 > $OUT/base.json 2> $OUT/base.err; show $OUT/base.json base
VARGENO_HIP_LIB=$R/variants/canon.so python3 bench.py $A --cpu-sample 1000 > $OUT/canon.json 2> $OUT/canon.err; show $OUT/canon.json canon; grep -E "parity|Error|assert" $OUT/canon.err | tee -a $OUT/summary.txt
python3 bench.py $A --cpu-sample 0 > $OUT/base2.json 2> $OUT/base2.err; show $OUT/base2.json base2
VARGENO_HIP_LIB=$R/variants/canon.so python3 bench.py $A --cpu-sample 0 > $OUT/canon2.json 2> $OUT/canon2.err; show $OUT/canon2.json canon2
( VARGENO_HIP_LIB=$R/variants/canon.so python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fastq.py -x -q -m gpu ) > $OUT/pytest_canon.log 2>&1; tail -3 $OUT/pytest_canon.log | tee -a $OUT/summary.txt
