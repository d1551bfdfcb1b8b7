#!/bin/bash
# Round 6, lease o: footprint against speed with the table-width steps (profiles/budget_sweep.py) on the shipped build.  (The pool hands the SAME box to consecutive
# leases: what earlier leases left in /tmp and /dev/shm -- 75 GB of a 79 GB root, 47 GB of tmpfs -- goes first.)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_o
mkdir -p $OUT
rm -rf /tmp/vg_bench /tmp/vg_bench_job /dev/shm/vg_bench* /tmp/pytest-of-* 2>/dev/null
df -h / /dev/shm | tail -2
cd $R
timeout 2400 python3 profiles/budget_sweep.py 0 235 205 192 175 128 96 > $OUT/budget_sweep_r06.json 2> $OUT/budget_sweep.err
cat $OUT/budget_sweep_r06.json | python3 -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    j=json.loads(l)
    print({k:(round(v,3) if isinstance(v,float) else v) for k,v in j.items() if k in ('budget_GB','device_GB','index_open_s','reads_per_s','ms_per_step','kernel','kernel_ms','frac','parity','failed')}, (j.get('plan') or '')[:330])
"
tail -3 $OUT/budget_sweep.err
