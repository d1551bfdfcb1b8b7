#!/bin/bash
# Round 6, lease x (library host code only; the kernels' machine code is unchanged: device_code_sha.sh first): smoke(), a short bench
# (parity of 200 000 reads with the oracle on the new library build), then the same-box A/B of vg_index_open's file ring.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_x
mkdir -p $OUT
rm -rf /tmp/vg_bench /tmp/vg_bench_job /dev/shm/vg_bench* /tmp/pytest-of-* 2>/dev/null
cd $R
bash profiles/device_code_sha.sh > $OUT/device_code.txt 2>&1; cat $OUT/device_code.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
( time timeout 600 python3 bench.py --gpus 1 --steps 5 --warmup 2 --secondary none --no-gather-probe --no-ingest --cpu-reference no --sustain-seconds 0 --cpu-sample 200000 --job-reads 0 ) > $OUT/bench.json 2> $OUT/bench.err
grep -E "^\[bench\] (parity|index resident|vg_index_open)" $OUT/bench.err | cut -c1-330
tail -c 300 $OUT/bench.json; echo
P=/tmp/vg_bench/g3100000000_s10000000_c24/idx
if [ -e $P.done ]; then
	timeout 700 python3 profiles/open_ab.py $P 20 p64x8:VG_FILE_PIECE_MB=64,VG_FILE_RING=8 p16x32:VG_FILE_PIECE_MB=16,VG_FILE_RING=32 p32x16:VG_FILE_PIECE_MB=32,VG_FILE_RING=16 p16x64:VG_FILE_PIECE_MB=16,VG_FILE_RING=64 p8x64:VG_FILE_PIECE_MB=8,VG_FILE_RING=64 p64x8_again:VG_FILE_PIECE_MB=64,VG_FILE_RING=8 p16x32_again:VG_FILE_PIECE_MB=16,VG_FILE_RING=32 > $OUT/open_ab.jsonl 2> $OUT/open_ab.err
	cat $OUT/open_ab.jsonl | cut -c1-900
	tail -3 $OUT/open_ab.err
fi
