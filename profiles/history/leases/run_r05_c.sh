#!/bin/bash
# Round 5, lease c: the one-block arena (vg_arena.h) + stage B1 with 32 signatures / 4 records per item.  (1) parity + budget + multi-replica tests on the new allocation path;
# (2) vg_index_open at hg38 scale with the arena and without (VG_NO_ARENA=1), phase by phase; (3) pack-kernel variants on the open index.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_c
mkdir -p $OUT
cd $R
( time timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q --durations=8 ) > $OUT/pytest.txt 2>&1
tail -16 $OUT/pytest.txt
B="--steps 20 --warmup 5 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --cpu-reference no --cpu-sample 0"
export VG_VERBOSE=1
line() { python3 - $1 $2 <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    o = j.get("other_input_form") or {}
    print("%-10s reads/s %.4g ms/step %.3f pack %.3f wave %.3f | gate words: ms/step %.3f pack %.3f wave %.3f | open %.2fs %.1f GB" % (sys.argv[2], j["value"], j["ms_per_step"], j["device_ms_per_step"]["pack"], j["device_ms_per_step"]["wave"],
          o.get("ms_per_step", 0), o.get("pack_ms", 0), o.get("wave_ms", 0), j["config"]["index_open_s"], j["config"]["index_bytes_hbm"] / 1e9))
except Exception as e:
    print(sys.argv[2], "failed", repr(e))
PY
}
run() { name=$1; shift; env "$@" timeout 600 python3 bench.py $B > $OUT/$name.json 2> $OUT/$name.err; line $OUT/$name.json $name; }
run arena VG_X=1
grep -h "vargeno_hip\] [a-zA-Z]" $OUT/arena.err | grep -v budget
run noarena VG_NO_ARENA=1
grep -h "vargeno_hip\] [a-zA-Z]" $OUT/noarena.err | grep -v budget
run arena2 VG_X=1
grep -h "vargeno_hip\] [a-zA-Z]" $OUT/arena2.err | grep -v budget
for v in packdry packg10 packg3 packnt0; do run $v VARGENO_HIP_LIB=$R/variants/$v.so; done
for b in 7 14 28; do run bpc$b VG_PACK_BPC=$b; done
run base_r04 VARGENO_HIP_LIB=$R/variants/base_r04.so
# the parity of the whole path once more at full scale, on the arena (8 M reads against the oracle)
timeout 900 python3 bench.py --steps 20 --warmup 5 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --cpu-reference no --cpu-sample 200000 > $OUT/parity.json 2> $OUT/parity.err
line $OUT/parity.json parity; grep -h "parity" $OUT/parity.err | tail -2
# the repeat-rich genome: stage B1's items (tree against the r04 library)
B2="--steps 20 --warmup 5 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --cpu-reference no --repeats 0.3"
timeout 900 python3 bench.py $B2 --cpu-sample 200000 > $OUT/rep_new.json 2> $OUT/rep_new.err; line $OUT/rep_new.json rep_new; grep -h "parity" $OUT/rep_new.err | tail -1
VARGENO_HIP_LIB=$R/variants/base_r04.so timeout 600 python3 bench.py $B2 --cpu-sample 0 > $OUT/rep_base.json 2> $OUT/rep_base.err; line $OUT/rep_base.json rep_base
timeout 600 python3 bench.py $B2 --cpu-sample 0 > $OUT/rep_new2.json 2> $OUT/rep_new2.err; line $OUT/rep_new2.json rep_new2
# the stress profile on the default index
B3="--steps 20 --warmup 5 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --cpu-reference no --lowq 0.5 --cpu-sample 0"
timeout 600 python3 bench.py $B3 > $OUT/lowq_new.json 2> $OUT/lowq_new.err; line $OUT/lowq_new.json lowq_new
VARGENO_HIP_LIB=$R/variants/base_r04.so timeout 600 python3 bench.py $B3 > $OUT/lowq_base.json 2> $OUT/lowq_base.err; line $OUT/lowq_base.json lowq_base
