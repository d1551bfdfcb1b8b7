#!/bin/bash
# Round 5, lease v: the bench's device pre-touch (a child takes the free memory once before the genome is generated) on a fresh box,
# main line only; then the eight-rank rehearsal with its per-replica start-up numbers printed.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_v
mkdir -p $OUT
cd $R
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --secondary none --cpu-reference no --no-ingest --job-reads 0 --cpu-sample 200000 --sustain-seconds 0 ) > $OUT/bench.json 2> $OUT/bench.err
grep "pre-touch\|taken once\|index resident\|phases" $OUT/bench.err | cut -c1-400
python3 -c "
import json
j=json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][-1])
print('value %.4g ms/step %.3f open %.2f pretouch %s' % (j['value'], j['ms_per_step'], j['config']['index_open_s'], j['config']['device_memory_pretouch']))"
( time timeout 900 python3 -m pytest tests/test_gpu_multi.py -m gpu -q -x -s -k "eight_ranks" ) > $OUT/eight.txt 2>&1
grep "eight replicas\|passed\|failed" $OUT/eight.txt | cut -c1-600
