#!/bin/bash
# Round 5: the driver's command on the tree (main line + job leg + secondary legs), as the driver runs it.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_bench
mkdir -p $OUT
cd $R
( time timeout 2400 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -40 $OUT/bench_default.err | cut -c1-300
python3 - $OUT/bench_default.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value %.4g ms/step %.3f frac %.3f open %.2f wall %.0f" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["config"]["index_open_s"], j["bench_wall_s"]))
print("job", json.dumps(j.get("job"))[:1500])
for k, v in (j.get("secondary") or {}).items():
    print(k, v.get("skipped") or "%.4g reads/s ms/step %.3f frac %.3f parity %s wall %.0f" % (v["value"], v["ms_per_step"], v["roofline"]["frac"], (v.get("parity") or {}).get("equal"), v.get("wall_s", 0)))
print("ingest", json.dumps({k: (v.get("value") if isinstance(v, dict) else v) for k, v in (j.get("ingest_end_to_end") or {}).get("paths", {}).items()}), (j.get("ingest_end_to_end") or {}).get("chosen"))
print("cpu", (j.get("cpu_baseline") or {}).get("value"), (j.get("cpu_baseline") or {}).get("kind"))
PY
