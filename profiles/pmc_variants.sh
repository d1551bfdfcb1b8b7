# one --pmc pass per prebuilt variant (variants/*.so): bash profiles/pmc_variants.sh "CTR1 CTR2 ..." [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
CTRS="$1"; shift
ARGS="${@:---steps 3 --warmup 1 --cpu-sample 0 --no-check --reads 8000000}"
cp $R/vargeno_amd/csrc/libvargeno_hip.so /tmp/shipped.so
cd /tmp && export TMPDIR=/tmp
for v in $R/variants/*.so; do
	name=$(basename $v .so)
	cp $v $R/vargeno_amd/csrc/libvargeno_hip.so
	rm -rf $R/gpurun_out/pmcv/$name
	rocprofv3 --pmc $CTRS --output-format csv -d $R/gpurun_out/pmcv/$name -- python3 $R/bench.py $ARGS > /dev/null 2>&1
done
cp /tmp/shipped.so $R/vargeno_amd/csrc/libvargeno_hip.so
python3 - <<PY
import csv,glob,collections,os
for d in sorted(glob.glob("$R/gpurun_out/pmcv/*")):
    for f in glob.glob(d+"/*/*_counter_collection.csv"):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)): agg[(r["Kernel_Name"][:52], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(agg.items()):
            if "wave_kernel<false, " in k[0] and ", 4>" in k[0]: print(os.path.basename(d), k[1], len(v), "%.5g"%(sum(v)/len(v)))
PY
