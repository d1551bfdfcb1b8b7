#!/bin/bash
# Round 5: where do the seconds of a COLD vg_index_open go (a fresh process, the device's memory just freed by another process or not)?
# The hg38-scale index is built once; then `vargeno geno` on a tiny FASTQ: (1) first process on the box, (2) right after (1) exits,
# (3) after 15 s of idle, (4) with the pre-packer off (VARGENO_PACK_THREADS=0), (5) right after a python process that held 250 GB exits.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_cold
mkdir -p $OUT
cd $R
python3 - <<'PY' > $OUT/build.log 2>&1
import sys, os
sys.path.insert(0, os.getcwd())
sys.argv = ["x"]
import bench
from vargeno_amd import synth
args = bench.parse_args()
g, s, _ = synth.genome_and_snps(genome_len=args.genome, n_snps=args.snps, n_chroms=args.chroms)
d = "/tmp/vg_bench/g%d_s%d_c%d" % (args.genome, args.snps, args.chroms)
bench.build_index_files(args, g, s, d, os.path.join(d, "idx"))
r = synth.make_reads(__import__("numpy").random.default_rng(1), g, s, 20000)
synth.write_fastq(os.path.join(d, "tiny.fq"), r)
PY
D=/tmp/vg_bench/g3100000000_s10000000_c24
cd $D
run() { name=$1; shift; ( time env VARGENO_VERBOSE=1 "$@" $R/vargeno_amd/csrc/vargeno geno idx tiny.fq snps.vcf $OUT/$name.vcf ) > $OUT/$name.log 2>&1; echo "== $name"; grep -h "index start-up\|^reads:\|^real" $OUT/$name.log | cut -c1-700; }
run first X=1
run second X=1
sleep 15
run after_idle X=1
run no_prepack VARGENO_PACK_THREADS=0
python3 -c "
import torch, time
t = torch.empty(250 * 2**30, dtype=torch.uint8, device='cuda'); t.fill_(1); torch.cuda.synchronize(); del t" 2>&1 | tail -1
run after_python_250GB X=1
