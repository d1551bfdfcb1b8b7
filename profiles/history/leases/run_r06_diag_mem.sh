#!/bin/bash
# diagnostic: who holds device memory while test_gpu_cli.py / test_gpu_fastq.py run
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/diag_mem
mkdir -p $OUT
rm -rf /tmp/pytest-of-* 2>/dev/null
cd $R
ls /sys/class/kfd/kfd/proc > $OUT/kfd_proc0.txt 2>&1
ps -eo pid,ppid,etimes,rss,args > $OUT/ps0.txt 2>&1
rocm-smi --showmeminfo vram --showpids > $OUT/smi0.txt 2>&1
( while true; do
    u=$(cat /sys/class/drm/card*/device/mem_info_vram_used 2>/dev/null | tr '\n' ' ')
    k=""
    for p in /sys/class/kfd/kfd/proc/*; do [ -d "$p" ] && k="$k $(basename $p):$(cat $p/vram_* 2>/dev/null | tr '\n' ',')"; done
    echo "$(date +%s.%N) vram_used $u kfd $k"
    sleep 0.25
  done ) > $OUT/vram.txt 2>&1 &
W=$!
( time timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_fastq.py -x -q -m gpu --durations=15 -p no:cacheprovider ) > $OUT/suite.txt 2>&1
echo "END $(date +%s.%N)" >> $OUT/suite.txt
sleep 2
kill $W
ps -eo pid,ppid,etimes,rss,args > $OUT/ps1.txt 2>&1
rocm-smi --showmeminfo vram --showpids > $OUT/smi1.txt 2>&1
tail -12 $OUT/suite.txt
head -c 600 $OUT/smi0.txt
awk '{print $3}' $OUT/vram.txt | sort -n | tail -1
