#!/bin/bash
# `vargeno geno` end to end at hg38 scale (BASELINE.json configs[2] index, 8 M reads = 2.5 GB of FASTQ): wall-time phases of the
# drop-in command line, device framing (default) and host framing.   bash profiles/run_cli_hg38_r02.sh  -> gpurun_out/cli_hg38/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cli_hg38
mkdir -p $OUT
cd $R
python3 - <<'PY' > $OUT/prep.log 2>&1
import os, subprocess, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from vargeno_amd import synth
d = os.environ.get("VG_BENCH_DIR", "/tmp/vg_bench") + "/g3100000000_s10000000_c24"
g, s, _ = synth.genome_and_snps(genome_len=3_100_000_000, n_snps=10_000_000, n_chroms=24)
if not os.path.exists(d + "/idx.done"):
    os.makedirs(d, exist_ok=True)
    synth.write_fasta(d + "/ref.fa", g); synth.write_vcf(d + "/snps.vcf", g, s)
    t0 = time.time()
    subprocess.check_call([os.path.join(os.getcwd(), "vargeno_amd/csrc/vargeno"), "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
    print("vargeno index: %.1f s" % (time.time() - t0))
    open(d + "/idx.done", "w").close()
src = synth.DeviceReadSource(g, s, torch.device("cuda", 0))
tb, tq, to = src.batch(31337, 8_000_000)
n, L = 8_000_000, 150
rec = 10 + 1 + L + 3 + L + 1
m = np.empty((n, rec), dtype=np.uint8)
ids = np.arange(n, dtype=np.int64)
m[:, 0] = ord("@"); m[:, 1] = ord("r")
for k in range(8):
    m[:, 2 + k] = 48 + (ids // 10 ** (7 - k)) % 10
m[:, 10] = 10
m[:, 11:11 + L] = tb.cpu().numpy().reshape(n, L)
m[:, 11 + L] = 10; m[:, 12 + L] = ord("+"); m[:, 13 + L] = 10
m[:, 14 + L:14 + 2 * L] = tq.cpu().numpy().reshape(n, L)
m[:, 14 + 2 * L] = 10
m.tofile(d + "/reads8m.fq")
print("FASTQ written:", os.path.getsize(d + "/reads8m.fq"))
PY
cat $OUT/prep.log | tail -3
D=${VG_BENCH_DIR:-/tmp/vg_bench}/g3100000000_s10000000_c24
for mode in 0 1; do
	( cd $D && time env VARGENO_VERBOSE=1 VG_VERBOSE=1 VARGENO_HOST_FASTQ=$mode $R/vargeno_amd/csrc/vargeno geno idx reads8m.fq snps.vcf out$mode.vcf ) > $OUT/geno_host$mode.log 2>&1
	grep -E "^reads:|^real|vargeno_hip" $OUT/geno_host$mode.log
done
cmp $D/out0.vcf $D/out1.vcf && echo "VCFs identical: $(grep -vc '^#' $D/out0.vcf) records" | tee -a $OUT/geno_host0.log
