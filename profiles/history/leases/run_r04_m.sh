#!/bin/bash
# Round 4, lease M: -DVG_RC_VOID=1 (a forward pass that had exact hits but was not processed, and whose buckets held no entry of the
# other strand, is not retried: the reverse-complement pass would find no exact hit) against the same sources without it (base0),
# chr22-scale both genomes + hg38 scale both genomes.  Variants built out of tree (the shipped sources are untouched).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_m
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
PART=${PART:-chr22}
if [ $PART = chr22 ]; then
for rep in 0 0.3; do
	tag=$( [ $rep = 0 ] && echo def || echo rep30 )
	VARGENO_HIP_LIB=$R/variants/base0.so c22 ${tag}_base --repeats $rep --cpu-sample 0
	VARGENO_HIP_LIB=$R/variants/rcvoid.so c22 ${tag}_rcvoid --repeats $rep
	VARGENO_HIP_LIB=$R/variants/rckeep.so c22 ${tag}_rckeep --repeats $rep
	VARGENO_HIP_LIB=$R/variants/base0.so c22 ${tag}_base2 --repeats $rep --cpu-sample 0
	VARGENO_HIP_LIB=$R/variants/rcvoid.so c22 ${tag}_rcvoid2 --repeats $rep --cpu-sample 0
	VARGENO_HIP_LIB=$R/variants/rckeep.so c22 ${tag}_rckeep2 --repeats $rep --cpu-sample 0
done
VARGENO_HIP_LIB=$R/variants/rckeep.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_rckeep.log 2>&1; tail -3 $OUT/pytest_rckeep.log | tee -a $OUT/summary.txt
else
h38() { local name=$1; shift; env "$@" timeout 900 python3 bench.py --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --steps 20 --warmup 5 $H38X > $OUT/$name.json 2> $OUT/$name.err; summ $name; }
h38 hg38_base VARGENO_HIP_LIB=$R/variants/base0.so
h38 hg38_$V VARGENO_HIP_LIB=$R/variants/$V.so
h38 hg38_base2 VARGENO_HIP_LIB=$R/variants/base0.so
h38 hg38_${V}2 VARGENO_HIP_LIB=$R/variants/$V.so
H38X="--repeats 0.3"
h38 hg38rep_base VARGENO_HIP_LIB=$R/variants/base0.so
H38X="--repeats 0.3 --cleanup"
h38 hg38rep_$V VARGENO_HIP_LIB=$R/variants/$V.so
fi
