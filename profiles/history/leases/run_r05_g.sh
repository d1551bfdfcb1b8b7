#!/bin/bash
# Round 5, lease g: auxiliary rows dealt to the lanes of the wave (all fetched in one wait, matched in the owners' order).  Parity tests, then hg38 scale: repeat-rich genome (parity on the
# 8 M-read batch + timing against the r04 library), default genome timing.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_g
mkdir -p $OUT
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x ) > $OUT/pytest.txt 2>&1
tail -6 $OUT/pytest.txt
line() { python3 - $1 $2 <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    o = j.get("other_input_form") or {}
    d = j["device_ms_per_step"]
    print("%-10s reads/s %.4g ms/step %.3f pack %.3f wave %.3f deep %.3f tail %.3f frac %.3f | gate words: ms/step %.3f | open %.2fs parity %s redone(timed build unknown) %s" % (sys.argv[2], j["value"], j["ms_per_step"], d["pack"], d["wave"], d["of_which_deep_list_wave_tier"], d["spill_tiers_overlapped"], j["roofline"]["frac"],
          o.get("ms_per_step", 0), j["config"]["index_open_s"], (j.get("parity") or {}).get("equal"), j.get("reads_per_step_redone_by_deep_list_tier")))
except Exception as e:
    print(sys.argv[2], "failed", repr(e))
PY
}
C="--steps 20 --warmup 5 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --cpu-reference no --job-reads 0"
timeout 900 python3 bench.py $C --repeats 0.3 --cpu-sample 200000 > $OUT/rep_new.json 2> $OUT/rep_new.err; line $OUT/rep_new.json rep_new; grep -h "Error" $OUT/rep_new.err | tail -2
VARGENO_HIP_LIB=$R/variants/base_r04.so timeout 600 python3 bench.py $C --repeats 0.3 --cpu-sample 0 > $OUT/rep_base.json 2> $OUT/rep_base.err; line $OUT/rep_base.json rep_base
timeout 600 python3 bench.py $C --repeats 0.3 --cpu-sample 0 > $OUT/rep_new2.json 2> $OUT/rep_new2.err; line $OUT/rep_new2.json rep_new2
timeout 600 python3 bench.py $C --cpu-sample 0 > $OUT/def_new.json 2> $OUT/def_new.err; line $OUT/def_new.json def_new
VARGENO_HIP_LIB=$R/variants/base_r04.so timeout 600 python3 bench.py $C --cpu-sample 0 > $OUT/def_base.json 2> $OUT/def_base.err; line $OUT/def_base.json def_base
VARGENO_HIP_LIB=$R/variants/clk.so timeout 600 python3 bench.py --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --cpu-reference no --job-reads 0 --repeats 0.3 --steps 1 --warmup 0 --cpu-sample 0 > $OUT/clk.json 2> $OUT/clk.err
grep -h "^CLK.*ecap 14" $OUT/clk.json $OUT/clk.err | sort | head -8
