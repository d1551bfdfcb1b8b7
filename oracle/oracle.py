"""ctypes wrapper around oracle/liboracle.so (the CPU restatement).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from
vargeno_amd/ (tests/test_layout.py enforces that).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
REF_BIN = os.path.join(HERE, "_ref", "vargeno")

STAT_FIELDS = ["reads", "reads_n", "reads_invalid", "passes", "passes_ok", "chunks", "gate_open",
               "refbf_pos", "snpbf_pos", "large_block", "ref_query", "snp_query", "ref_probe", "snp_probe",
               "scan_ref", "scan_snp", "scan_oob", "aux_ref", "aux_snp", "site_test", "ctx", "walks", "incr",
               "ingest_bytes"]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in STAT_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in STAT_FIELDS}


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(
            os.path.getmtime(os.path.join(HERE, f)) for f in ("vg_oracle.c", "vg_oracle.h")):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        u64p, u32p, u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
        L.vgo_index_from_arrays.restype = C.c_void_p
        L.vgo_index_from_arrays.argtypes = [C.c_uint64, u64p, u32p, u8p, C.c_uint64, u32p,
                                            C.c_uint64, u64p, u32p, u8p, u8p, u8p, u8p,
                                            C.c_uint64, u32p, u8p, C.c_uint64, u64p, C.c_uint64, u64p]
        L.vgo_index_load.restype = C.c_void_p
        L.vgo_index_load.argtypes = [C.c_char_p]
        L.vgo_index_free.argtypes = [C.c_void_p]
        L.vgo_set_scan_stride.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.vgo_reset_counts.argtypes = [C.c_void_p]
        L.vgo_process.restype = C.c_int64
        L.vgo_process.argtypes = [C.c_void_p, u8p, u8p, u64p, C.c_uint64, C.c_int, C.POINTER(Stats)]
        L.vgo_num_sites.restype = C.c_uint64
        L.vgo_num_sites.argtypes = [C.c_void_p]
        L.vgo_get_sites.argtypes = [C.c_void_p, u32p, u8p, u8p, u8p, u8p, u8p, u8p]
        L.vgo_call.restype = C.c_int
        L.vgo_call.argtypes = [C.c_int, C.c_int, C.c_uint8, C.c_uint8, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.vgo_trace_read.argtypes = [C.c_void_p, u8p, u8p, C.c_uint64, u32p]
        L.vgo_vote_replay.argtypes = [u32p, u32p, u32p, C.c_uint64, u32p]
        L.vgo_alg_bytes.restype = C.c_uint64
        L.vgo_alg_bytes.argtypes = [C.POINTER(Stats)]
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


class OracleIndex:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle index construction failed")
        self.h = C.c_void_p(handle)
        self.stats = Stats()

    @classmethod
    def load(cls, prefix):
        return cls(lib().vgo_index_load(os.fsencode(prefix)))

    @classmethod
    def from_arrays(cls, a):
        """a: dict of numpy arrays with the keys of vargeno_amd.index_io.IndexArrays."""
        c = lambda k, dt: np.ascontiguousarray(a[k], dtype=dt)
        rk, rp, ra, rx = c("ref_kmer", np.uint64), c("ref_pos", np.uint32), c("ref_amb", np.uint8), c("ref_aux", np.uint32)
        sk, sp, si, sa = c("snp_kmer", np.uint64), c("snp_pos", np.uint32), c("snp_info", np.uint8), c("snp_amb", np.uint8)
        srf, saf = c("snp_rf", np.uint8), c("snp_af", np.uint8)
        sxp, sxi = c("snp_aux_pos", np.uint32), c("snp_aux_info", np.uint8)
        rw, sw = c("ref_bf_words", np.uint64), c("snp_bf_words", np.uint64)
        h = lib().vgo_index_from_arrays(
            len(rk), _p(rk, C.c_uint64), _p(rp, C.c_uint32), _p(ra, C.c_uint8), rx.size // 10, _p(rx, C.c_uint32),
            len(sk), _p(sk, C.c_uint64), _p(sp, C.c_uint32), _p(si, C.c_uint8), _p(sa, C.c_uint8), _p(srf, C.c_uint8),
            _p(saf, C.c_uint8), sxp.size // 10, _p(sxp, C.c_uint32), _p(sxi, C.c_uint8),
            int(a["ref_bf_bits"]), _p(rw, C.c_uint64), int(a["snp_bf_bits"]), _p(sw, C.c_uint64))
        return cls(h)

    def close(self):
        if self.h:
            lib().vgo_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scan_stride(self, r, s):
        lib().vgo_set_scan_stride(self.h, r, s)

    def reset(self):
        lib().vgo_reset_counts(self.h)
        self.stats = Stats()

    def process(self, bases, quals, offsets, nthreads=1):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        return int(lib().vgo_process(self.h, _p(bases, C.c_uint8), _p(quals, C.c_uint8), _p(offsets, C.c_uint64),
                                     len(offsets) - 1, nthreads, C.byref(self.stats)))

    def sites(self):
        n = int(lib().vgo_num_sites(self.h))
        pos = np.empty(n, np.uint32)
        u8 = [np.empty(n, np.uint8) for _ in range(6)]
        lib().vgo_get_sites(self.h, _p(pos, C.c_uint32), *[_p(x, C.c_uint8) for x in u8])
        return dict(pos=pos, ref_base=u8[0], alt_base=u8[1], ref_freq=u8[2], alt_freq=u8[3], ref_cnt=u8[4], alt_cnt=u8[5])

    def trace(self, bases, quals):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        out = np.zeros(5, np.uint32)
        lib().vgo_trace_read(self.h, _p(bases, C.c_uint8), _p(quals, C.c_uint8), len(bases), _p(out, C.c_uint32))
        return out

    def alg_bytes(self):
        return int(lib().vgo_alg_bytes(C.byref(self.stats)))


def vote_replay(index, kpos, neigh):
    """The oracle's vote state machine alone on a sequence of adds -> (has_best, best index, best freq, ambiguous)."""
    index = np.ascontiguousarray(index, dtype=np.uint32); kpos = np.ascontiguousarray(kpos, dtype=np.uint32); neigh = np.ascontiguousarray(neigh, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().vgo_vote_replay(_p(index, C.c_uint32), _p(kpos, C.c_uint32), _p(neigh, C.c_uint32), len(index), _p(out, C.c_uint32))
    return tuple(int(x) for x in out)


def call(ref_cnt, alt_cnt, ref_freq, alt_freq):
    conf, gq = C.c_double(), C.c_int()
    g = lib().vgo_call(int(ref_cnt), int(alt_cnt), int(ref_freq), int(alt_freq), C.byref(conf), C.byref(gq))
    return g, conf.value, gq.value


GT_STR = {1: "0/0", 2: "1/1", 3: "0/1"}


def calls_by_key(sites, chrlens):
    """{'chrname$pos': (GT string, GQ)} the way qv.cc:1573-1626 keys its map.  chrlens: [(name, len)]."""
    out = {}
    for i in range(len(sites["pos"])):
        g, conf, gq = call(sites["ref_cnt"][i], sites["alt_cnt"][i], sites["ref_freq"][i], sites["alt_freq"][i])
        if g == 0:
            continue
        idx = int(sites["pos"][i])
        j = 0
        while j < len(chrlens) and idx > chrlens[j][1]:
            idx -= chrlens[j][1]
            j += 1
        out["%s$%d" % (chrlens[j][0], idx)] = (GT_STR[g], gq)
    return out


def read_chrlens(path):
    out = []
    with open(path) as f:
        for line in f:
            name, ln = line.split()[:2]
            out.append((name[:32], int(ln)))
    return out


def parse_vcf_calls(path_or_text):
    """Genotyped records of a `vargeno geno` output VCF -> {'chr$pos': (GT, GQ)} (GT:GQ are the last two FORMAT keys)."""
    if "\n" in path_or_text:
        lines = path_or_text.splitlines()
    else:
        import gzip
        op = gzip.open if path_or_text.endswith(".gz") else open
        with op(path_or_text, "rt") as f:
            lines = f.read().splitlines()
    out = {}
    for ln in lines:
        if not ln or ln[0] == "#":
            continue
        c = ln.split("\t")
        chrom = c[0] if c[0].startswith("c") else "chr" + c[0]
        fmt = c[8].split(":")
        val = c[9].split(":")
        out["%s$%s" % (chrom, c[1])] = (val[fmt.index("GT")], int(val[fmt.index("GQ")]))
    return out
