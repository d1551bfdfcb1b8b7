#!/bin/bash
# Round 6, lease f: the SNP dictionary's LO32-ordered view, same-box A/B (VG_NO_SSEC=1 = the round-5 way: high-half SNP queries one by one):
# default workload + lowq50 + len250 on the open index, repeats30 as a child leg.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_f
mkdir -p $OUT
cd $R
for v in ssec nossec; do
	unset VG_NO_SSEC; [ $v = nossec ] && export VG_NO_SSEC=1
	timeout 1500 python3 bench.py --steps 20 --warmup 5 --secondary lowq50,len250,repeats30 --job-reads 0 --no-ingest --cpu-reference no --sustain-seconds 0 --no-gather-probe > $OUT/bench_$v.json 2> $OUT/bench_$v.err
	python3 - $OUT/bench_$v.json $v <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "default: value %.4g ms/step %.3f kernel %.3f pack %.3f frac %.3f hbm %.1f GB parity %s" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["device_ms_per_step"]["pack"], j["roofline"]["frac"], j["config"]["index_bytes_hbm"] / 1e9, (j.get("parity") or {}).get("equal")))
for k, v in (j.get("secondary") or {}).items():
    print(sys.argv[2], k, v.get("skipped") or "value %.4g ms/step %.3f kernel %.3f frac %.3f parity %s" % (v["value"], v["ms_per_step"], v["kernel_ms"], v["frac"], v["parity"]))
PY
done
unset VG_NO_SSEC
cp /tmp/vg_bench/bench_detail_g3100000000_s10000000_c24.json $OUT/ 2>/dev/null
