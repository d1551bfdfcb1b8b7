#!/bin/bash
# A/B of prebuilt library variants (variants/*.so, built here with different -D knobs) on one GPU box:
#   bash profiles/ab_variants.sh [bench args]     -> gpurun_out/ab_<name>.json
# The shipped library is restored at the end.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cp vargeno_amd/csrc/libvargeno_hip.so /tmp/shipped.so
for v in variants/*.so; do
	name=$(basename "$v" .so)
	cp "$v" vargeno_amd/csrc/libvargeno_hip.so
	for rep in 1; do
		timeout 600 python3 bench.py "$@" 2>gpurun_out/ab_$name.err | tail -1 > gpurun_out/ab_${name}_$rep.json
		python3 - "$name" "$rep" <<'PY'
import json, sys
try:
    j = json.load(open("gpurun_out/ab_%s_%s.json" % (sys.argv[1], sys.argv[2])))
    print(sys.argv[1], sys.argv[2], "pack %.4f" % j["device_ms_per_step"]["pack"], "reads/s %.4g" % j["value"], "ms/step %.4f" % j["ms_per_step"], "wave ms %.4f" % j["roofline"].get("kernel_ms", -1), "frac %.3f" % j["roofline"]["frac"], "spilled", j.get("reads_per_step_spilled_to_lane_tier"), "tiers %.3f" % j["device_ms_per_step"]["spill_tiers_overlapped"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
	done
done
cp /tmp/shipped.so vargeno_amd/csrc/libvargeno_hip.so
