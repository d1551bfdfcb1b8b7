#!/bin/bash
# hg38-scale start-up: phases of vg_index_open (VG_VERBOSE=1) through `vargeno geno` on an empty FASTQ
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/vg_hg38_load
mkdir -p $D; cd $D
python3 - <<PY
import sys; sys.path.insert(0, "$R")
from vargeno_amd import synth
g, s, r = synth.chr22_scale(genome_len=3_100_000_000, n_snps=10_000_000, n_reads=1000, n_chroms=24)
synth.write_fasta("ref.fa", g); synth.write_vcf("snps.vcf", g, s)
open("empty.fq", "w").close()
PY
VARGENO_NO_LITE=1 $R/vargeno_amd/csrc/vargeno index ref.fa snps.vcf idx > /dev/null 2>&1
for rep in 1 2; do VG_VERBOSE=1 VARGENO_VERBOSE=1 $R/vargeno_amd/csrc/vargeno geno idx empty.fq snps.vcf out.vcf 2>&1 | grep -E "vargeno_hip|reads:"; echo; done
