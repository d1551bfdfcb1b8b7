#!/bin/bash
# BASELINE.json configs[4]'s index (hg38 + ~100 M SNPs) on one replica, round 3: the -m gpu test of that layout, then bench.py on
# the same index files -- the shipped layout (paired HI32 table), the round-2 layout (VG_NO_HX=1), stage clocks, and the
# rocprofv3 kernel trace + L2 / EA counters of vg_wave_kernel_big.   -> gpurun_out/hg38f_r03/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/hg38f_r03
mkdir -p $OUT
cd $R
export VG_BENCH_DIR=/dev/shm/vg_bench VARGENO_VERBOSE=1 VG_VERBOSE=1
if [ -z "$SKIP_TEST" ]; then
	( time timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k hg38f -s ) > $OUT/pytest_hg38f.log 2>&1
	tail -5 $OUT/pytest_hg38f.log
fi
A="--workload hg38f --no-ingest --no-gather-probe --steps 10 --warmup 2"
( time timeout 1500 python3 bench.py $A --cpu-sample 500000 ) > $OUT/bench.json 2> $OUT/bench.err
tail -8 $OUT/bench.err
VG_NO_HX=1 timeout 900 python3 bench.py $A --cpu-sample 0 > $OUT/bench_nohx.json 2> $OUT/bench_nohx.err
for v in $R/variants/*.so; do
	n=$(basename $v .so)
	VARGENO_HIP_LIB=$v timeout 900 python3 bench.py $A --cpu-sample 0 --steps $([ $n = clk ] && echo 1 || echo 10) --warmup $([ $n = clk ] && echo 0 || echo 2) > $OUT/v_$n.json 2> $OUT/v_$n.err
done
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py $A --cpu-sample 0"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $CMD > $OUT/kt.json 2> $OUT/kt.err
timeout 900 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $CMD > $OUT/pmc_l2.json 2> $OUT/pmc_l2.err
timeout 900 rocprofv3 --pmc TCC_EA0_RDREQ_128B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_DRAM_32B --output-format csv -d $OUT/pmc_ea -- $CMD > $OUT/pmc_ea.json 2> $OUT/pmc_ea.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.json 2> $OUT/pmc_write.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
python3 $R/profiles/summarize_prof.py $OUT r03_hg38f "vg_wave_kernel_big<" > /dev/null
rm -rf $OUT/kt/*/*_kernel_trace.csv $OUT/*/*/*agent_info.csv
rm -rf /dev/shm/vg_bench
for f in $OUT/bench.json $OUT/bench_nohx.json $OUT/v_*.json; do python3 - $f <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); d = j["device_ms_per_step"]
    print("%-40s reads/s %.4g  ms/step %.3f  wave %.3f  pack %.3f  tiers %.3f  frac %.3f  views %s" % (sys.argv[1].split("/")[-1], j["value"], j["ms_per_step"], d["wave"], d["pack"], d["spill_tiers_overlapped"], j["roofline"]["frac"], j["config"].get("index_views")))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
grep -h "CLK" $OUT/v_clk.json | head -3
