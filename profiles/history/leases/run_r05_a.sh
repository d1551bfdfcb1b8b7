#!/bin/bash
# Round 5, lease a: (1) what allocation costs (tools/alloc_probe); (2) parity of the position-parallel pack kernel (GPU parity + FASTQ tests);
# (3) same-lease A/B at hg38 scale: the r04 library (variants/base_r04.so) against the tree, VG_VERBOSE phase times of vg_index_open.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_a
mkdir -p $OUT
cd $R
timeout 300 vargeno_amd/csrc/tools/alloc_probe 64 > $OUT/alloc_probe.jsonl 2>&1
cat $OUT/alloc_probe.jsonl | cut -c1-300
( time timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fastq.py -m gpu -x -q ) > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
B="--steps 20 --warmup 5 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --cpu-reference no"
export VG_VERBOSE=1
for w in chr22 hg38; do
	timeout 900 python3 bench.py --workload $w $B --cpu-sample 200000 > $OUT/new_$w.json 2> $OUT/new_$w.err
	VARGENO_HIP_LIB=$R/variants/base_r04.so timeout 600 python3 bench.py --workload $w $B --cpu-sample 0 > $OUT/base_$w.json 2> $OUT/base_$w.err
	timeout 600 python3 bench.py --workload $w $B --cpu-sample 0 > $OUT/new2_$w.json 2> $OUT/new2_$w.err
	for t in new base new2; do python3 - $OUT/${t}_$w.json $t $w <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    o = j.get("other_input_form") or {}
    print("%-5s %-5s reads/s %.4g ms/step %.3f pack %.3f wave %.3f | gate words: ms/step %.3f pack %.3f wave %.3f | open %.1fs parity %s" % (sys.argv[2], sys.argv[3], j["value"], j["ms_per_step"], j["device_ms_per_step"]["pack"], j["device_ms_per_step"]["wave"],
          o.get("ms_per_step", 0), o.get("pack_ms", 0), o.get("wave_ms", 0), j["config"]["index_open_s"], (j.get("parity") or {}).get("equal")))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "failed", repr(e))
PY
	done
	grep -h "vargeno_hip\] [a-zA-Z]" $OUT/new2_$w.err | grep -v budget
done
