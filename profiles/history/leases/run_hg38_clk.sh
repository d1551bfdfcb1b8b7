#!/bin/bash
# hg38-scale: per-stage cycle shares of the wave kernel (dbg/clk.so = the library built with -DVG_STAGE_CLOCKS; dev aid)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/hg38
export VG_BENCH_DIR=/tmp/vg_bench_hg38
cp $R/vargeno_amd/csrc/libvargeno_hip.so /tmp/shipped.so
cp $R/dbg/clk.so $R/vargeno_amd/csrc/libvargeno_hip.so
VG_NO_PACK_OVERLAP=1 python3 $R/bench.py --workload hg38 --cpu-sample 0 --no-check --steps 1 --warmup 0 > $R/gpurun_out/hg38/clk_hg38.txt 2> $R/gpurun_out/hg38/clk_hg38.err
cp /tmp/shipped.so $R/vargeno_amd/csrc/libvargeno_hip.so
tail -3 $R/gpurun_out/hg38/clk_hg38.err
grep -c CLK $R/gpurun_out/hg38/clk_hg38.txt
