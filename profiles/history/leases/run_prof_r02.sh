#!/bin/bash
# Round-2 profiling recipe for the DEFAULT bench (hg38-scale, BASELINE.json configs[2]); run on the MI355X box from the repo
# root: `bash profiles/run_prof_r02.sh [tag]`.  Kernel-trace/--stats and each --pmc group are separate rocprofv3 runs of the
# SAME bench.py command (program directly after `--`).  Raw output -> gpurun_out/prof_<tag>/ (scratch);
# profiles/summarize_prof.py condenses it into summary_<tag>.txt / traffic_<tag>.json, which are copied into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VARGENO_VERBOSE=1 VG_VERBOSE=1
CMD="python3 $R/bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-gather-probe --no-ingest"
( time python3 $R/bench.py --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $CMD > $OUT/kt.json 2> $OUT/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $CMD > $OUT/pmc_l2.json 2> $OUT/pmc_l2.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
python3 $R/profiles/summarize_prof.py $OUT $TAG > /dev/null
# per-stage cycle shares and list-overflow reasons (development build with -DVG_STAGE_CLOCKS, if it came along)
if [ -f $R/variants/clk.so ]; then
	VARGENO_HIP_LIB=$R/variants/clk.so python3 $R/bench.py --cpu-sample 0 --no-gather-probe --no-ingest --steps 1 --warmup 0 > $OUT/clk.txt 2> $OUT/clk.err
fi
# keep the merge small: the raw traces stay on the box
rm -rf $OUT/kt/*/*_kernel_trace.csv $OUT/kt/*/*agent_info.csv
du -sh $OUT
tail -3 $OUT/bench_default.err
