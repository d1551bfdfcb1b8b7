#!/bin/bash
# Round 5, lease r: same-lease A/B of the four-stream layout (deep wave tiers on `tail`, lane tiers + counter copies on `tail2`)
# against variants/pretail.so (one tail stream): default workload, chr22, 250 bp reads; then the parity tests on the shipped library.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_r
mkdir -p $OUT
cd $R
COMMON="--secondary none --cpu-sample 0 --no-gather-probe --no-ingest --sustain-seconds 0 --job-reads 0"
show() { python3 - $1 "$2" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
d = j["device_ms_per_step"]
print("%-18s %.4g reads/s ms/step %.3f pack %.3f wave %.3f tail %.3f (deep %.3f) open %.2f" % (sys.argv[2], j["value"], j["ms_per_step"], d["pack"], d["wave"], d["spill_tiers_overlapped"], d["of_which_deep_list_wave_tier"], j["config"]["index_open_s"]))
PY
}
for round in 1 2; do
	for lib in shipped pretail; do
		if [ $lib = pretail ]; then export VARGENO_HIP_LIB=$R/variants/pretail.so; else unset VARGENO_HIP_LIB; fi
		sleep 25
		timeout 600 python3 bench.py $COMMON --steps 20 --warmup 5 > $OUT/default_${lib}_$round.json 2> $OUT/default_${lib}_$round.err; show $OUT/default_${lib}_$round.json default_${lib}_$round
		sleep 25
		timeout 600 python3 bench.py --workload chr22 --steps 40 --warmup 5 $COMMON > $OUT/chr22_${lib}_$round.json 2> $OUT/chr22_${lib}_$round.err; show $OUT/chr22_${lib}_$round.json chr22_${lib}_$round
		if [ $round = 1 ]; then
			sleep 25
			timeout 600 python3 bench.py --read-len 250 $COMMON --steps 20 --warmup 3 > $OUT/len250_${lib}.json 2> $OUT/len250_${lib}.err; show $OUT/len250_${lib}.json len250_${lib}
		fi
	done
done
unset VARGENO_HIP_LIB
( time timeout 1200 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not hg38" ) > $OUT/pytest.txt 2>&1
grep "passed\|failed" $OUT/pytest.txt | tail -2
