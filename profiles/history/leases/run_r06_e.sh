#!/bin/bash
# Round 6, lease e: the view census (variants/census.so = -DVG_VIEW_COUNTERS) on the default and the repeat-rich hg38-scale genome, 1 M reads each.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_e
mkdir -p $OUT
cd $R
for g in default repeats30; do
	EXTRA=""; [ $g = repeats30 ] && EXTRA="--repeats 0.3"
	VARGENO_HIP_LIB=$R/variants/census.so timeout 1500 python3 bench.py --reads 1000000 --steps 1 --warmup 0 --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --job-reads 0 --no-pretouch $EXTRA > $OUT/census_$g.json 2> $OUT/census_$g.err
	grep "view census" $OUT/census_$g.err > $OUT/census_$g.txt
	echo "== $g"; cat $OUT/census_$g.txt | cut -c1-150
	# the same launch on the shipped build: kernel time and events per read
	timeout 1500 python3 bench.py --reads 1000000 --steps 3 --warmup 1 --cpu-sample 0 --no-gather-probe --no-ingest --secondary none --sustain-seconds 0 --job-reads 0 --no-pretouch $EXTRA > $OUT/ship_$g.json 2> $OUT/ship_$g.err
	python3 - $OUT/ship_$g.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
d = json.load(open(j["detail"]))
print("shipped build: kernel %.4f ms per 1 M reads, alg bytes/read %.1f, events/read %s" % (j["roofline"]["kernel_ms"], j["roofline"]["algorithmic_bytes_per_read"], json.dumps({k: round(v, 3) for k, v in d["events_per_read"].items()})))
PY
done
