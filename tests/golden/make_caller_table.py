#!/usr/bin/env python3
"""tests/golden/caller_table.npz from the REAL reference's genotype caller (choose_best_genotype, src/qv.cc:1789-1848) evaluated
over its whole domain by oracle/ref_caller_table.cc (`make -C oracle caller_table`, build container only): for 16 encoded
allele-frequency pairs, every (ref_cnt, alt_cnt) in [0, 63]^2 -> genotype code, GQ as the VCF pass derives it (qv.cc:1681),
and the confidence itself.  The fixture is data (inputs and outputs of the reference function)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "caller_table"], stdout=subprocess.DEVNULL)
    rec = np.dtype([("ref_freq", "u1"), ("alt_freq", "u1"), ("ref_cnt", "u1"), ("alt_cnt", "u1"), ("genotype", "<i4"), ("gq", "<i4"), ("conf", "<f8")])
    t = np.fromfile(os.path.join(ROOT, "oracle", "_ref", "caller_table.bin"), dtype=rec)
    assert len(t) == 16 * 64 * 64
    np.savez_compressed(os.path.join(OUT, "caller_table.npz"), **{k: t[k] for k in rec.names})
    print("%d entries; genotypes: %s" % (len(t), dict(zip(*np.unique(t["genotype"], return_counts=True)))))
