#!/bin/bash
# Round 5, lease l: the command line after its pre-packer got a piece-maker thread and a parallel copy (FASTQ / CLI tests), the job
# leg alone, then the footprint-against-speed sweep (profiles/budget_sweep.py) on the same index files.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_l
mkdir -p $OUT
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_cli.py tests/test_gpu_fastq.py -m gpu -q -x -k "not hg38" ) > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
bash profiles/run_r05_job.sh 200000000
( time timeout 1500 python3 profiles/budget_sweep.py ) > $OUT/budget_sweep.jsonl 2> $OUT/budget_sweep.err
tail -3 $OUT/budget_sweep.err
python3 - $OUT/budget_sweep.jsonl <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if not ln.startswith("{"): continue
    j = json.loads(ln)
    if "failed" in j: print(j); continue
    print("budget %s: device %.1f GB, views %s, open %.2f s, %.4g reads/s, ms/step %.3f, kernel %.3f ms (%s), frac %.3f" % (j["budget_GB"], j["device_GB"], ",".join(j["views"]), j["index_open_s"], j["reads_per_s"], j["ms_per_step"], j["kernel_ms"], j["kernel"], j["roofline_frac"]))
PY
