#!/bin/bash
# end-to-end `vargeno geno` wall time on the chr22-scale set, device vs host FASTQ framing (index load included)
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/vg_bench/g40000000_s1000000_c1
python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2>&1     # builds the index files
python3 - <<PY
import sys; sys.path.insert(0, "$R")
from vargeno_amd import synth
g, s, r = synth.chr22_scale(n_reads=4000000)
synth.write_fastq("$D/reads4m.fq", r)
PY
ls -la $D/reads4m.fq
cd $D
for mode in 0 1; do
  for rep in 1 2; do
    /usr/bin/env VARGENO_HOST_FASTQ=$mode VARGENO_VERBOSE=1 $R/vargeno_amd/csrc/vargeno geno idx reads4m.fq snps.vcf out_$mode.vcf 2>&1 | grep -E "reads:"
  done
done
cmp out_0.vcf out_1.vcf && echo SAME_VCF
