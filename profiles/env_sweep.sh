# bash profiles/env_sweep.sh VAR v1 v2 ... -- [bench args]   : default bench under each value of one environment knob
VAR=$1; shift
vals=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do vals+=("$1"); shift; done
shift
mkdir -p gpurun_out
for v in "${vals[@]}"; do
	env $VAR=$v python3 bench.py --cpu-sample 0 --no-check "$@" 2>/dev/null | tail -1 > gpurun_out/sweep.json
	python3 -c "
import json
j=json.load(open('gpurun_out/sweep.json')); print('$VAR=$v', '%.4g'%j['value'], '%.4f'%j['ms_per_step'], j['device_ms_per_step'])"
done
