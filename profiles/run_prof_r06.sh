#!/bin/bash
# Round-6 profiling recipe; run on the MI355X box from the repo root:  bash profiles/run_prof_r06.sh <tag> [bench.py arguments of the workload]
#   bash profiles/run_prof_r06.sh r06                         the default bench (hg38-scale, BASELINE.json configs[2])
#   bash profiles/run_prof_r06.sh r06_repeats30 --repeats 0.3
#   bash profiles/run_prof_r06.sh r06_chr22 --workload chr22 --steps 40
# The kernel trace and each --pmc group are separate rocprofv3 runs of the SAME bench.py command (program directly after `--`;
# counters never together with a trace).  Raw output -> gpurun_out/prof_<tag>/ (scratch); profiles/summarize_prof.py condenses
# it into summary_<tag>.txt / traffic_<tag>.json, which are copied into profiles/ (rocprof_summary_<tag>.txt, traffic_<tag>.json).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VARGENO_VERBOSE=1 VG_VERBOSE=1
KSEL="vg_wave_kernel<false, "
case "$*" in *hg38f*) KSEL="vg_wave_kernel_big<";; esac
CMD="python3 $R/bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --job-reads 0 --no-pretouch $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $CMD > $OUT/kt.json 2> $OUT/kt.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $CMD > $OUT/pmc_l2.json 2> $OUT/pmc_l2.err
rocprofv3 --pmc TCC_EA0_RDREQ_128B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_DRAM_32B --output-format csv -d $OUT/pmc_ea -- $CMD > $OUT/pmc_ea.json 2> $OUT/pmc_ea.err
python3 $R/profiles/summarize_prof.py $OUT $TAG "$KSEL" > /dev/null
# (gpurun copies at most 64 MiB back: the raw per-dispatch CSVs -- 14 MB per workload -- stay on the box, the summaries are what is judged)
rm -rf $OUT/kt $OUT/pmc_write $OUT/pmc_l2 $OUT/pmc_ea
for f in $OUT/*.err; do tail -c 4000 $f > $f.tail; rm -f $f; done
du -sh $OUT
cp $OUT/summary_$TAG.txt $R/gpurun_out/rocprof_summary_$TAG.txt; cp $OUT/traffic_$TAG.json $R/gpurun_out/ 2>/dev/null
