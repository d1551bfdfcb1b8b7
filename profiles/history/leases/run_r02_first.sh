cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
nproc > gpurun_out/r2a/host.txt; free -g >> gpurun_out/r2a/host.txt; df -h /tmp >> gpurun_out/r2a/host.txt
( time python bench.py --workload chr22 --steps 20 --warmup 3 ) > gpurun_out/r2a/bench_chr22.json 2> gpurun_out/r2a/bench_chr22.err
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r2a/pytest.log 2>&1
tail -5 gpurun_out/r2a/pytest.log
