#!/bin/bash
# Round 5, lease q: same-lease A/B of the tail-stream change: the shipped library against variants/pretail.so (commit fdc3942: one tail
# stream for everything), default workload and chr22, alternating, the device idle for 25 s before each run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_q
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
	done
done
unset VARGENO_HIP_LIB
sleep 25
VG_PACK_OVERLAP=0 timeout 600 python3 bench.py --workload chr22 --steps 40 --warmup 5 $COMMON > $OUT/chr22_ov0.json 2> $OUT/chr22_ov0.err; show $OUT/chr22_ov0.json chr22_shipped_ov0
