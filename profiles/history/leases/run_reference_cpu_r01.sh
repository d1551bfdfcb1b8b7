#!/bin/bash
# The REFERENCE binary (oracle/_ref/vargeno, built in the build container from /root/reference by oracle/Makefile) timed on
# the GPU box's own host: `geno` on the chr22-scale set with 1 M reads and with an empty FASTQ; the difference of the two
# wall times is its read loop (the rest is its 16 GiB jump-table fill and file parsing).  Also compares its VCF with ours.
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/vg_bench/g40000000_s1000000_c1
python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2>&1     # builds the index files with the product's `vargeno index`
python3 - <<PY
import sys; sys.path.insert(0, "$R")
from vargeno_amd import synth
g, s, r = synth.chr22_scale(n_reads=1000000)
synth.write_fastq("$D/reads1m.fq", r)
open("$D/empty.fq", "w").close()
PY
cd $D
REF=$R/oracle/_ref/vargeno
t() { local s=$(date +%s.%N); "$@" > /dev/null 2>&1; local e=$(date +%s.%N); python3 -c "print('%.2f' % ($e - $s))"; }
for rep in 1 2; do
  T0=$(t $REF geno idx empty.fq snps.vcf ref_empty.vcf)
  T1=$(t $REF geno idx reads1m.fq snps.vcf ref_out.vcf)
  python3 -c "print('reference geno wall: empty FASTQ $T0 s, 1 M reads $T1 s -> read loop %.2f s = %.3g reads/s (1 thread)' % ($T1 - $T0, 1e6 / ($T1 - $T0)))"
done
$R/vargeno_amd/csrc/vargeno geno idx reads1m.fq snps.vcf ours.vcf > /dev/null 2>&1
cmp ref_out.vcf ours.vcf && echo "VCF of the reference and of the HIP path: byte-identical ($(grep -vc '^#' ours.vcf) records)"
nproc; grep -m1 "model name" /proc/cpuinfo
