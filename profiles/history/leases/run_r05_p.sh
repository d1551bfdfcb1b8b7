#!/bin/bash
# Round 5, lease p: deep wave tiers on one stream again, a lane-tier stream + scratch per batch slot -- the parity tests (every tier, both kernels), then
# 250 bp reads (their batches' tiers took 13 ms on the one tail stream), the default workload and chr22.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_p
mkdir -p $OUT
cd $R
( time timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -q -x -k "not hg38" ) > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
COMMON="--secondary none --cpu-sample 0 --no-gather-probe --no-ingest --sustain-seconds 0 --job-reads 0"
show() { python3 - $1 "$2" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
d = j["device_ms_per_step"]
print("%-10s %.4g reads/s ms/step %.3f pack %.3f wave %.3f tail %.3f (deep %.3f) deep reads %s lane reads %s open %.2f" % (sys.argv[2], j["value"], j["ms_per_step"], d["pack"], d["wave"], d["spill_tiers_overlapped"], d["of_which_deep_list_wave_tier"], j["reads_per_step_redone_by_deep_list_tier"], j["reads_per_step_sent_on_to_lane_tier"], j["config"]["index_open_s"]))
PY
}
timeout 600 python3 bench.py --read-len 250 $COMMON --steps 20 --warmup 3 > $OUT/len250.json 2> $OUT/len250.err; show $OUT/len250.json len250
timeout 600 python3 bench.py $COMMON --steps 20 --warmup 5 > $OUT/default.json 2> $OUT/default.err; show $OUT/default.json default
timeout 600 python3 bench.py --lowq 0.5 $COMMON --steps 20 --warmup 5 > $OUT/lowq50.json 2> $OUT/lowq50.err; show $OUT/lowq50.json lowq50
timeout 600 python3 bench.py --workload chr22 --steps 40 --warmup 5 $COMMON > $OUT/chr22.json 2> $OUT/chr22.err; show $OUT/chr22.json chr22
timeout 900 python3 bench.py --repeats 0.3 $COMMON --steps 20 --warmup 5 > $OUT/repeats30.json 2> $OUT/repeats30.err; show $OUT/repeats30.json repeats30
VG_PACK_OVERLAP=0 timeout 600 python3 bench.py --workload chr22 --steps 40 --warmup 5 $COMMON > $OUT/chr22_ov0.json 2> $OUT/chr22_ov0.err; show $OUT/chr22_ov0.json chr22_ov0
