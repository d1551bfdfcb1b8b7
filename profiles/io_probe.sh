#!/bin/bash
# Development aid: how fast can `vargeno index` write its dictionary on this box, per write mode and file system?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/io_probe
mkdir -p $OUT
{
df -h / /tmp /dev/shm 2>&1
mount | grep -E " on (/|/tmp|/dev/shm) " 
python3 - <<PY
import sys, time
sys.path.insert(0, "$R")
from vargeno_amd import synth
t = time.time()
g, s, _ = synth.genome_and_snps(genome_len=400_000_000, n_snps=1_300_000, n_chroms=4)
for d in ("/tmp/ioprobe", "/dev/shm/ioprobe"):
    import os
    try:
        os.makedirs(d, exist_ok=True)
        synth.write_fasta(d + "/ref.fa", g); synth.write_vcf(d + "/snps.vcf", g, s)
    except Exception as e:
        print("cannot use", d, e)
print("inputs: %.1f s" % (time.time() - t))
PY
for d in /tmp/ioprobe /dev/shm/ioprobe; do
	[ -f $d/ref.fa ] || continue
	for m in stream pwrite mmap; do
		echo "== $d $m"
		( cd $d && VARGENO_WRITE_MODE=$m VARGENO_VERBOSE=1 VARGENO_NO_LITE=1 timeout 300 $R/vargeno_amd/csrc/vargeno index ref.fa snps.vcf idx 2>&1 >/dev/null | grep -E "written|filled|sorted|partitioned" )
	done
	rm -rf $d
done
} > $OUT/io_probe.txt 2>&1
cat $OUT/io_probe.txt
