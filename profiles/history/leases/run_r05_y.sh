#!/bin/bash
# Round 5, lease y (after the quiet second in front of the timed region): why the bench's chr22 child leg measures 0.39-0.41 ms per step where the same workload stand-alone measures 0.35-0.36:
# the child leg's own command, then the same without the oracle legs, then the first again.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_y
mkdir -p $OUT
cd $R
show() { python3 - $1 "$2" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
d = j["device_ms_per_step"]
print("%-22s %.4g reads/s ms/step %.3f pack %.3f wave %.3f tail %.3f | gate words %.3f" % (sys.argv[2], j["value"], j["ms_per_step"], d["pack"], d["wave"], d["spill_tiers_overlapped"], (j.get("other_input_form") or {}).get("ms_per_step", 0)))
PY
}
CHILD="--gpus 1 --secondary none --no-gather-probe --no-ingest --cpu-reference no --sustain-seconds 0 --job-reads 0 --workload chr22 --steps 40 --warmup 5"
for v in "a_child --cpu-sample 200000" "b_nosample --cpu-sample 0" "c_child --cpu-sample 200000"; do
	set -- $v
	tag=$1; shift
	sleep 20
	timeout 600 python3 bench.py $CHILD "$@" > $OUT/$tag.json 2> $OUT/$tag.err; show $OUT/$tag.json $tag
done
