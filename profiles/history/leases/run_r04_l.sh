#!/bin/bash
# Round 4, lease L: a read's k-mers kept in registers (-DVG_KMER_REGS=1) against the shipped build, chr22-scale both genomes + hg38 scale.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_l
mkdir -p $OUT
cd $R
summ() {
	python3 - $OUT/$1.json $1 <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    d = j["device_ms_per_step"]
    o = j["other_input_form"]
    print("%-16s reads/s %.4g  ms/step %.3f  wave %.3f  pack %.3f  tiers %.3f  frac %.3f  spilled %s | gate words: %.4g  ms/step %.3f wave %.3f pack %.3f" % (sys.argv[2], j["value"], j["ms_per_step"], d["wave"], d["pack"], d["spill_tiers_overlapped"], j["roofline"]["frac"], j.get("reads_per_step_redone_by_deep_list_tier"), o["value"], o["ms_per_step"], o["wave_ms"], o["pack_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
c22() { local name=$1; shift; timeout 600 python3 bench.py --workload chr22 --cpu-reference no --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --steps 40 --warmup 5 "$@" > $OUT/$name.json 2> $OUT/$name.err; summ $name; grep parity $OUT/$name.err | tee -a $OUT/summary.txt; }
for rep in 0 0.3; do
	tag=$( [ $rep = 0 ] && echo def || echo rep30 )
	c22 ${tag}_base --repeats $rep --cpu-sample 0
	VARGENO_HIP_LIB=$R/variants/kreg.so c22 ${tag}_kreg --repeats $rep
	c22 ${tag}_base2 --repeats $rep --cpu-sample 0
	VARGENO_HIP_LIB=$R/variants/kreg.so c22 ${tag}_kreg2 --repeats $rep --cpu-sample 0
done
VARGENO_HIP_LIB=$R/variants/kreg.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_kreg.log 2>&1; tail -3 $OUT/pytest_kreg.log | tee -a $OUT/summary.txt
h38() { local name=$1; shift; env "$@" timeout 900 python3 bench.py --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --steps 20 --warmup 5 > $OUT/$name.json 2> $OUT/$name.err; summ $name; }
h38 hg38_base X=1
h38 hg38_kreg VARGENO_HIP_LIB=$R/variants/kreg.so
h38 hg38_base2 X=1
h38 hg38_kreg2 VARGENO_HIP_LIB=$R/variants/kreg.so
