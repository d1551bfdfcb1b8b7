#!/usr/bin/env python3
"""One-off differential run on the MI355X (profiles/fuzz_lowcomplex_r03.txt): synth.f_lowcomplex for as many seeds as fit
in the time given, index by `vargeno index`, HIP path (counting and timed build) against the oracle; every third seed on a
fall-back layout; with a third argument "stress" the reads have 90 % gate-open chunks, 3 % errors, lengths 33-300 and lower-case
bases (profiles/fuzz_lowcomplex_stress_r03.txt).  The -m gpu suite keeps seeds 1-6 (tests/test_gpu_parity.py::test_low_complexity_genomes)."""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from oracle import oracle as O  # noqa: E402
from test_gpu_parity import CMP_STATS  # noqa: E402
from vargeno_amd import synth  # noqa: E402
from vargeno_amd.api import GenoIndex  # noqa: E402

BIN = os.path.join(ROOT, "vargeno_amd", "csrc", "vargeno")
first, budget = int(sys.argv[1]), float(sys.argv[2])
STRESS = len(sys.argv) > 3 and sys.argv[3] == "stress"      # reads with 90 % gate-open chunks, 3 % errors, lengths 33-300, lower case
t0 = time.time()
seed, fails = first, 0
KNOBS = (None, None, "VG_NO_MX", None, None, "VG_NO_DIRECT", None, None, "VG_NO_MX+VG_NO_HX")
while time.time() - t0 < budget:
    knob = KNOBS[seed % len(KNOBS)]
    g, s, r = synth.f_lowcomplex(seed, n_reads=10 if STRESS else 6_000)
    if STRESS:
        r = synth.make_reads(np.random.default_rng(1000 + seed), g, s, 6000, lengths=(150, 101, 64, 33, 250, 300), err=0.03, lowq=0.9, lower_frac=0.02)
    d = tempfile.mkdtemp(prefix="lc%d_" % seed)
    synth.write_fasta(os.path.join(d, "ref.fa"), g)
    synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
    ox = O.OracleIndex.load(os.path.join(d, "idx"))
    ox.process(r.bases, r.quals, r.offsets, nthreads=8)
    so, want = ox.sites(), ox.stats.as_dict()
    for k in ("VG_NO_MX", "VG_NO_DIRECT", "VG_NO_HX"):
        os.environ.pop(k, None)
    for k in (knob.split("+") if knob else ()):
        os.environ[k] = "1"
    ok = True
    with GenoIndex.open(os.path.join(d, "idx")) as gx:
        for stats in (True, False):
            gx.reset()
            gx.set_stats(stats)
            gx.submit(r.bases, r.quals, r.offsets)
            rc, ac = gx.counts()
            ok = ok and np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"])
            if stats:
                st = gx.stats()
                ok = ok and all(st[k] == want[k] for k in CMP_STATS)
    fails += not ok
    print("seed %3d  %-18s genome %d  sites %d  counted %d  aux_ref %d  walks %d  %s" % (seed, knob or "shipped layout", g.total_len, len(so["pos"]), int(so["ref_cnt"].sum() + so["alt_cnt"].sum()), want["aux_ref"], want["walks"], "identical" if ok else "DIFFERENT"), flush=True)
    shutil.rmtree(d)
    seed += 1
print("%d seeds (%d-%d), %d different, %.0f s" % (seed - first, first, seed - 1, fails, time.time() - t0))
sys.exit(1 if fails else 0)
