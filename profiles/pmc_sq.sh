# instruction mix and issue-cycle counters of the wave kernel (separate --pmc passes; no trace domains)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-sample 0 --no-check --reads 8000000"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $R/gpurun_out/sq/p1 -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/sq/p2 -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/sq/p3 -- python3 $R/bench.py $ARGS > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
for P in ["p1","p2","p3"]:
    for f in glob.glob("$R/gpurun_out/sq/%s/*/*_counter_collection.csv"%P):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)): agg[(r["Kernel_Name"][:52], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(agg.items()):
            if "wave_kernel<false, " in k[0] and ", 4>" in k[0]: print(P, k[1], len(v), "%.5g"%(sum(v)/len(v)))
PY
