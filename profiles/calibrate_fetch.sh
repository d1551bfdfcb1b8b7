#!/bin/bash
# What does FETCH_SIZE count for one random 8-byte gather?  (MI355X_MICROARCH.md: calibrate on a known count in
# your own access pattern.)  gather_probe issues a known number of gathers per kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/cal
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/cal/f -- $R/vargeno_amd/csrc/tools/gather_probe 16 > $R/gpurun_out/cal/probe.out 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/cal/r -- $R/vargeno_amd/csrc/tools/gather_probe 16 > /dev/null 2>&1
python3 - <<PY
import csv,glob
for g in ["f","r"]:
    for f in glob.glob("$R/gpurun_out/cal/%s/*/*_counter_collection.csv"%g):
        rows=list(csv.DictReader(open(f)))
        for r in rows[:9]: print(r["Kernel_Name"][:34], "grid", r["Grid_Size"], r["Counter_Name"], r["Counter_Value"])
PY
head -3 $R/gpurun_out/cal/probe.out
