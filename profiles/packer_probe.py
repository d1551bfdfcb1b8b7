#!/usr/bin/env python3
"""Development aid (no device): throughput of the library's host-side FASTQ framing + 2-bit packing (vg_packer_*) by thread count.
   python3 profiles/packer_probe.py [reads] [threads ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import numpy as np

from vargeno_amd._lib import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
threads = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
L = 150
rec = 10 + 1 + L + 3 + L + 1
rng = np.random.default_rng(1)
m = np.empty((n, rec), np.uint8)
ids = np.arange(n, dtype=np.int64)
m[:, 0] = ord("@"); m[:, 1] = ord("r")
for k in range(8):
    m[:, 2 + k] = 48 + (ids // 10 ** (7 - k)) % 10
m[:, 10] = 10
m[:, 11:11 + L] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=(n, L))]
m[:, 11 + L] = 10; m[:, 12 + L] = ord("+"); m[:, 13 + L] = 10
m[:, 14 + L:14 + 2 * L] = rng.integers(ord("#"), ord("J"), size=(n, L), dtype=np.uint8)
m[:, 14 + 2 * L] = 10
text = m.reshape(-1)
Lb = lib()
rc_, kc_ = int(Lb.vg_packer_reads_cap(len(text))), int(Lb.vg_packer_kmers_cap(len(text)))
kmers, meta, offs = np.zeros(kc_, np.uint64), np.zeros(rc_, np.uint64), np.zeros(rc_ + 1, np.uint64)
print("host: %d hardware threads; affinity %d; cgroup cpu.max %s" % (os.cpu_count(), len(os.sched_getaffinity(0)), (open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")))
for t in threads:
    h = C.c_void_p()
    assert Lb.vg_packer_create(t, C.byref(h)) == 0
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    best = 1e9
    for rep in range(5):
        Lb.vg_packer_begin(h)
        t0 = time.perf_counter()
        assert Lb.vg_packer_push(h, text.ctypes.data_as(C.c_void_p), len(text), kmers.ctypes.data_as(C.c_void_p), kc_, meta.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), rc_, C.byref(a), C.byref(b), C.byref(c)) == 0
        best = min(best, time.perf_counter() - t0)
    assert a.value == n
    Lb.vg_packer_destroy(h)
    print("threads %3d: %.1f ms per %d reads = %.3g reads/s = %.1f GB/s of text" % (t, 1e3 * best, n, n / best, len(text) / best / 1e9), flush=True)
