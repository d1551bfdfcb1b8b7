R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
unset VG_NO_DIRECT
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/cmp2/l2_direct -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-sample 0 --reads 8000000 > /dev/null 2>&1
export VG_NO_DIRECT=1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/cmp2/l2_jump -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-sample 0 --reads 8000000 > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
for V in ["direct","jump"]:
    for f in glob.glob("$R/gpurun_out/cmp2/l2_%s/*/*_counter_collection.csv"%V):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)): agg[(r["Kernel_Name"][:52], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(agg.items()):
            if "wave_kernel<false, " in k[0] and ", 4>" in k[0]: print(V, k[1], len(v), "%.4g"%(sum(v)/len(v)))
PY
