mkdir -p gpurun_out/prof
python bench.py --steps 10 --warmup 2 --reads 8000000 --cpu-sample 0 > gpurun_out/bench_8m.json 2> gpurun_out/bench_8m.err
cat gpurun_out/bench_8m.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/kt -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-sample 0 > $R/gpurun_out/prof/kt.json 2> $R/gpurun_out/prof/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/prof/pmc_fetch.json 2> $R/gpurun_out/prof/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/prof/pmc_write.json 2> $R/gpurun_out/prof/pmc_write.err
cd $R/gpurun_out/prof && find . -type f | head -30 && du -sh .
