#!/bin/bash
# Round 5, lease n: the pre-packer reading through a mapping (FASTQ / CLI tests, the job leg alone); why 250 bp reads leave a 14 ms
# tail per batch (the stage-clock variant prints each tier's overflow reasons); three hg38-scale replicas opening at once (gloo).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_n
mkdir -p $OUT
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_cli.py tests/test_gpu_fastq.py -m gpu -q -x -k "not hg38" ) > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
bash profiles/run_r05_job.sh 200000000
COMMON="--secondary none --cpu-sample 0 --no-gather-probe --no-ingest --sustain-seconds 0 --job-reads 0"
VARGENO_HIP_LIB=$R/variants/clk.so timeout 600 python3 bench.py --read-len 250 $COMMON --steps 5 --warmup 2 > $OUT/len250_clk.json 2> $OUT/len250_clk.err
grep "dbg\|cycles" $OUT/len250_clk.err | tail -30 | cut -c1-300
timeout 600 python3 bench.py --read-len 250 $COMMON --steps 10 --warmup 2 > $OUT/len250.json 2> $OUT/len250.err
python3 - $OUT/len250.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("len250: %.4g reads/s ms/step %.3f" % (j["value"], j["ms_per_step"]), j["device_ms_per_step"]["pack"], j["device_ms_per_step"]["wave"], "tail", j["device_ms_per_step"]["spill_tiers_overlapped"], j["device_ms_per_step"]["of_which_deep_list_wave_tier"], "deep", j["reads_per_step_redone_by_deep_list_tier"], "lane", j["reads_per_step_sent_on_to_lane_tier"])
PY
( time VG_BENCH_BACKEND=gloo timeout 900 python3 bench.py --gpus 3 --reads 2000000 --batches 2 --steps 5 --warmup 2 $COMMON ) > $OUT/three_replicas.json 2> $OUT/three_replicas.err
tail -5 $OUT/three_replicas.err | cut -c1-300
python3 - $OUT/three_replicas.json <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    pr = j["multi_gpu_per_rank"]
    print("three replicas on one device: open wall", pr["index_open_s"], "cpu", pr["index_open_cpu_s"], "GB", pr["index_device_GB"], "value %.4g" % j["value"])
    print(j["config"]["index_plan"][:300])
except Exception as e:
    print("three replicas: no line", e)
PY
