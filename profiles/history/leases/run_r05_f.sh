#!/bin/bash
# Round 5, lease f: the repeat-rich genome at hg38 scale -- parity of the tree on the 8 M-read batch (the extra passes of stage B1), then
# stage clocks of the tree and of round 4's kernel (-DVG_STAGE_CLOCKS builds) on one 8 M-read step each.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_f
mkdir -p $OUT
cd $R
B2="--no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --cpu-reference no --repeats 0.3 --job-reads 0"
timeout 900 python3 bench.py $B2 --steps 20 --warmup 5 --cpu-sample 200000 > $OUT/rep_new.json 2> $OUT/rep_new.err
grep -h "parity\|Error\|error" $OUT/rep_new.err | tail -3
python3 -c "
import json; j=json.loads([l for l in open('$OUT/rep_new.json') if l.startswith('{')][-1]); d=j['device_ms_per_step']; print('rep_new ms/step %.3f pack %.3f wave %.3f deep %.3f frac %.3f redone %s' % (j['ms_per_step'], d['pack'], d['wave'], d['of_which_deep_list_wave_tier'], j['roofline']['frac'], j['reads_per_step_redone_by_deep_list_tier']))"
for v in clk clk_r04; do
	VARGENO_HIP_LIB=$R/variants/$v.so timeout 600 python3 bench.py $B2 --steps 1 --warmup 0 --cpu-sample 0 > $OUT/$v.json 2> $OUT/$v.err
	echo "== $v"; grep -h "^CLK\|^DBG\|^\[dbg\]" $OUT/$v.json $OUT/$v.err | sort | head -60
done
