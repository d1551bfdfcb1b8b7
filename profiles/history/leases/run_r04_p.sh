#!/bin/bash
# Round 4, lease P: the second half of the round's kernel work in the tree (rows against the whole key table, key filter before the site bytes in stage B1,
# rows that repeat a position): the parity tests, the new fixture through the CLI against the reference's VCF, and the shipped library against base0 (the kernels before).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_p
mkdir -p $OUT
cd $R
( time timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py::test_cli_vcf_is_byte_identical_to_the_reference tests/test_vote_aggregate.py -x -q -m gpu --durations=8 ) > $OUT/pytest.log 2>&1
tail -14 $OUT/pytest.log
c22() { local name=$1; shift; timeout 600 python3 bench.py --workload chr22 --cpu-reference no --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --steps 40 --warmup 5 "$@" > $OUT/$name.json 2> $OUT/$name.err; python3 - $OUT/$name.json $name <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); d = j["device_ms_per_step"]
print("%-14s reads/s %.4g ms/step %.3f wave %.3f pack %.3f frac %.3f spilled %s" % (sys.argv[2], j["value"], j["ms_per_step"], d["wave"], d["pack"], j["roofline"]["frac"], j.get("reads_per_step_redone_by_deep_list_tier")))
PY
grep parity $OUT/$name.err; }
VARGENO_HIP_LIB=$R/variants/base0.so c22 rep30_base --repeats 0.3 --cpu-sample 0
c22 rep30_tree --repeats 0.3
c22 def_tree --repeats 0
