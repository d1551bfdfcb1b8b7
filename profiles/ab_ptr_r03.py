#!/usr/bin/env python3
"""Development aid (no torch): A/B of library builds on a chr22-scale index through the C-ABI's host-buffer entry point.
   python3 profiles/ab_ptr_r03.py <workdir> <lib.so> [<lib.so> ...]      ("-" = the shipped library)
Synthesises 40 Mbp / 1 M SNPs / 1 M reads, indexes them with `vargeno index`, runs the oracle, then for every library in a child
process: open, timed build, one warm-up batch + 6 batches of the same 1 M reads; prints the main-tier kernel time per batch
(HIP events inside the library) and whether the counters equal the oracle's."""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

if sys.argv[1] == "--child":
    from vargeno_amd.api import GenoIndex

    d = sys.argv[2]
    b, q, o = (np.load(os.path.join(d, n + ".npy")) for n in ("bases", "quals", "offsets"))
    t0 = time.time()
    with GenoIndex.open(os.path.join(d, "idx")) as gx:
        t_open = time.time() - t0
        gx.set_stats(False)
        gx.submit(b, q, o)
        rc, ac = gx.counts()
        h = hashlib.sha256(rc.tobytes() + ac.tobytes()).hexdigest()[:16]
        gx.timing()                                   # (averages over the batches since the previous call)
        per = []
        for _ in range(6):
            gx.reset()
            gx.submit(b, q, o)
            gx.counts()
            per.append(gx.timing()["ms_main"])
        print("open %.1f s  counters %s  main-tier kernel ms per 1 M-read batch: %s  (min %.4f)" % (t_open, h, " ".join("%.4f" % x for x in per), min(per)), flush=True)
    sys.exit(0)

from oracle import oracle as O  # noqa: E402
from vargeno_amd import synth  # noqa: E402

d = sys.argv[1]
os.makedirs(d, exist_ok=True)
t0 = time.time()
g, s, rng = synth.genome_and_snps()
r = synth.make_reads(rng, g, s, 1_000_000)
synth.write_fasta(os.path.join(d, "ref.fa"), g)
synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
for n, a in (("bases", r.bases), ("quals", r.quals), ("offsets", r.offsets)):
    np.save(os.path.join(d, n + ".npy"), a)
subprocess.check_call([os.path.join(ROOT, "vargeno_amd", "csrc", "vargeno"), "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
print("inputs + index %.1f s" % (time.time() - t0), flush=True)
t0 = time.time()
ox = O.OracleIndex.load(os.path.join(d, "idx"))
ox.process(r.bases, r.quals, r.offsets, nthreads=os.cpu_count())
so = ox.sites()
print("oracle %.1f s  counters %s" % (time.time() - t0, hashlib.sha256(so["ref_cnt"].tobytes() + so["alt_cnt"].tobytes()).hexdigest()[:16]), flush=True)
for lib in sys.argv[2:]:
    env = dict(os.environ)
    if lib != "-":
        env["VARGENO_HIP_LIB"] = os.path.abspath(lib)
    print("%-28s " % lib, end="", flush=True)
    subprocess.call([sys.executable, os.path.abspath(__file__), "--child", d], env=env)
