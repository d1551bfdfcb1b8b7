#!/bin/bash
# hg38-scale, live: the reference binary (oracle/_ref/vargeno) and the product's `vargeno geno` (HIP path) on the same index
# files (written by the product's `vargeno index`) and the same 2 M reads; the two VCFs are compared byte for byte.
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/vg_hg38_ref
mkdir -p $D $R/gpurun_out/hg38
free -g | head -2
cd $D
python3 - <<PY
import sys, time; sys.path.insert(0, "$R")
from vargeno_amd import synth
t = time.time()
g, s, r = synth.chr22_scale(genome_len=3_100_000_000, n_snps=10_000_000, n_reads=2_000_000, n_chroms=24)
synth.write_fasta("ref.fa", g); synth.write_vcf("snps.vcf", g, s); synth.write_fastq("reads.fq", r)
print("inputs written in %.0f s" % (time.time() - t))
PY
( time VARGENO_NO_LITE=1 $R/vargeno_amd/csrc/vargeno index ref.fa snps.vcf idx ) 2>&1 | grep -E "real" | sed 's/^/vargeno index (product): /'
( time VARGENO_VERBOSE=1 $R/vargeno_amd/csrc/vargeno geno idx reads.fq snps.vcf ours.vcf ) 2>&1 | grep -E "reads:|real" | sed 's/^/product geno: /'
( time timeout 2400 $R/oracle/_ref/vargeno geno idx reads.fq snps.vcf ref.vcf ) 2>&1 | grep -E "Time|real|rror|Abort|Segm" | sed 's/^/reference geno: /'
ls -la ours.vcf ref.vcf
cmp ours.vcf ref.vcf && echo "hg38-scale: VCF of the reference and of the HIP path byte-identical, $(grep -vc '^#' ours.vcf) genotyped records"
