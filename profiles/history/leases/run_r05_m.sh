#!/bin/bash
# Round 5, lease m: the read store (packed batches parked in device memory while the index opens) -- FASTQ / CLI / budget tests,
# the job leg alone, the footprint-against-speed sweep with the block holding permanent arrays only when views are left out, and
# chr22 with the pack kernel of batch k+1 under batch k's wave kernel (VG_PACK_OVERLAP) against without.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_m
mkdir -p $OUT
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_cli.py tests/test_gpu_fastq.py tests/test_gpu_parity.py -m gpu -q -x -k "not hg38 and (cli or fastq or store or packed or budget or framing)" ) > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
bash profiles/run_r05_job.sh 200000000
( time timeout 1500 python3 profiles/budget_sweep.py ) > $OUT/budget_sweep.jsonl 2> $OUT/budget_sweep.err
tail -3 $OUT/budget_sweep.err
python3 - $OUT/budget_sweep.jsonl <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if not ln.startswith("{"): continue
    j = json.loads(ln)
    if "failed" in j: print(str(j)[:300]); continue
    print("budget %s: device %.1f GB, views %s, open %.2f s, %.4g reads/s, ms/step %.3f, kernel %.3f ms (%s), frac %.3f | %s" % (j["budget_GB"], j["device_GB"], ",".join(j["views"]), j["index_open_s"], j["reads_per_s"], j["ms_per_step"], j["kernel_ms"], j["kernel"], j["roofline_frac"], j.get("memory")))
PY
for ov in 0 1; do
	VG_PACK_OVERLAP=$ov python3 bench.py --workload chr22 --steps 40 --warmup 5 --secondary none --no-gather-probe --no-ingest --cpu-reference no --sustain-seconds 0 --cpu-sample 0 --job-reads 0 > $OUT/chr22_ov$ov.json 2> $OUT/chr22_ov$ov.err
	python3 -c "
import json,sys
j=json.loads([l for l in open('$OUT/chr22_ov$ov.json') if l.startswith('{')][-1])
print('chr22 overlap $ov: %.4g reads/s ms/step %.3f pack %.3f wave %.3f' % (j['value'], j['ms_per_step'], j['device_ms_per_step']['pack'], j['device_ms_per_step']['wave']))"
done
