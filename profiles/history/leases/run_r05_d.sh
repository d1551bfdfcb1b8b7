#!/bin/bash
# Round 5, lease d: (1) stage clocks (-DVG_STAGE_CLOCKS builds: variants/clk_r04.so = round 4's kernel, variants/clk.so = the tree) on
# chr22-scale indexes, default and repeat-rich genome: where did the repeat-rich launch lose what the default one gained?
# (2) the whole GPU suite + smoke on the tree.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_d
mkdir -p $OUT
cd $R
B="--workload chr22 --steps 1 --warmup 0 --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --cpu-reference no --job-reads 0"
for rep in 0 0.3; do
	for v in clk_r04 clk; do
		VARGENO_HIP_LIB=$R/variants/$v.so timeout 600 python3 bench.py $B --repeats $rep > $OUT/${v}_$rep.json 2> $OUT/${v}_$rep.err
		echo "== $v repeats $rep"; grep -h "^CLK\|^DBG" $OUT/${v}_$rep.json $OUT/${v}_$rep.err | head -12
	done
done
B2="--workload chr22 --steps 20 --warmup 5 --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --cpu-reference no --job-reads 0 --repeats 0.3"
for v in base_r04; do VARGENO_HIP_LIB=$R/variants/$v.so timeout 600 python3 bench.py $B2 2> $OUT/t_$v.err | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['ms_per_step'], j['device_ms_per_step'])"; done
timeout 600 python3 bench.py $B2 2> $OUT/t_new.err | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('new', j['ms_per_step'], j['device_ms_per_step'])"
( time timeout 1500 python3 -m pytest tests -m gpu -q --durations=15 ) > $OUT/pytest.txt 2>&1
tail -30 $OUT/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
