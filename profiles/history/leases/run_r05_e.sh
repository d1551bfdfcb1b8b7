#!/bin/bash
# Round 5, lease e: the whole GPU suite + smoke on the tree; then the repeat-rich genome at hg38 scale: the tree (items of four LO32 records whose
# further neighbours get passes of their own) against the r04 library and against variants with one record / eight signatures per item.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_e
mkdir -p $OUT
cd $R
( time timeout 1500 python3 -m pytest tests -m gpu -q -x --durations=15 ) > $OUT/pytest.txt 2>&1
tail -28 $OUT/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
line() { python3 - $1 $2 <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    o = j.get("other_input_form") or {}
    d = j["device_ms_per_step"]
    print("%-10s reads/s %.4g ms/step %.3f pack %.3f wave %.3f deep %.3f tail %.3f | gate words: ms/step %.3f | open %.2fs parity %s" % (sys.argv[2], j["value"], j["ms_per_step"], d["pack"], d["wave"], d["of_which_deep_list_wave_tier"], d["spill_tiers_overlapped"],
          o.get("ms_per_step", 0), j["config"]["index_open_s"], (j.get("parity") or {}).get("equal")))
except Exception as e:
    print(sys.argv[2], "failed", repr(e))
PY
}
B2="--steps 20 --warmup 5 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --cpu-reference no --repeats 0.3 --job-reads 0"
timeout 900 python3 bench.py $B2 --cpu-sample 200000 > $OUT/rep_new.json 2> $OUT/rep_new.err; line $OUT/rep_new.json rep_new
for v in base_r04 lw0 lw1 sig1; do VARGENO_HIP_LIB=$R/variants/$v.so timeout 600 python3 bench.py $B2 --cpu-sample 0 > $OUT/rep_$v.json 2> $OUT/rep_$v.err; line $OUT/rep_$v.json rep_$v; done
timeout 600 python3 bench.py $B2 --cpu-sample 0 > $OUT/rep_new2.json 2> $OUT/rep_new2.err; line $OUT/rep_new2.json rep_new2
