"""numpy readers/writers for the reference's on-disk index: plumbing for tests and
bench.py.  The product's own readers are in csrc/ (C++); these exist so that Python drivers can hand
the same arrays to vg_index_create and so tests can inspect files.

  <p>.ref.dict  u64 n, u64 n_aux, n x {u64 kmer, u32 pos, u8 ambig}, n_aux x 10 x u32        (dictgen.c:63-154)
  <p>.snp.dict  u64 n, u64 n_aux, n x {u64 kmer, u32 pos, u8 snp, u8 ambig, u8 rf, u8 af},
                n_aux x {u64 kmer, 10 x {u32 pos, u8 snp, u8 rf, u8 af}}                      (dictgen.c:156-275)
  <p>.*.bf      u64 bit count, ceil(bits/64) u64 words                                        (sdsl int_vector.hpp:1563-1595)
  <p>.chrlens   "name len\\n" per sequence                                                     (qv.cc:2343-2345)
"""
from __future__ import annotations

import os

import numpy as np

REF_DT = np.dtype([("kmer", "<u8"), ("pos", "<u4"), ("amb", "u1")])
SNP_DT = np.dtype([("kmer", "<u8"), ("pos", "<u4"), ("info", "u1"), ("amb", "u1"), ("rf", "u1"), ("af", "u1")])
SNP_AUX_DT = np.dtype([("kmer", "<u8"), ("e", np.dtype([("pos", "<u4"), ("info", "u1"), ("rf", "u1"), ("af", "u1")]), (10,))])
assert REF_DT.itemsize == 13 and SNP_DT.itemsize == 16 and SNP_AUX_DT.itemsize == 78

REF_BF_BITS = 1_200_000_000 * 8      # generate_bf.h:199 REF_BF_BYTES
REF_LITE_BF_BITS = 2_300_000_000 * 8
SNP_BF_BITS = 140_000_000 * 8


def read_ref_dict(path):
    with open(path, "rb") as f:
        n, n_aux = np.fromfile(f, "<u8", 2)
        rec = np.fromfile(f, REF_DT, int(n))
        aux = np.fromfile(f, "<u4", int(n_aux) * 10)
    return dict(ref_kmer=np.ascontiguousarray(rec["kmer"]), ref_pos=np.ascontiguousarray(rec["pos"]),
                ref_amb=np.ascontiguousarray(rec["amb"]), ref_aux=aux)


def read_snp_dict(path):
    with open(path, "rb") as f:
        n, n_aux = np.fromfile(f, "<u8", 2)
        rec = np.fromfile(f, SNP_DT, int(n))
        aux = np.fromfile(f, SNP_AUX_DT, int(n_aux))
    return dict(snp_kmer=np.ascontiguousarray(rec["kmer"]), snp_pos=np.ascontiguousarray(rec["pos"]),
                snp_info=np.ascontiguousarray(rec["info"]), snp_amb=np.ascontiguousarray(rec["amb"]),
                snp_rf=np.ascontiguousarray(rec["rf"]), snp_af=np.ascontiguousarray(rec["af"]),
                snp_aux_pos=np.ascontiguousarray(aux["e"]["pos"]).reshape(-1),
                snp_aux_info=np.ascontiguousarray(aux["e"]["info"]).reshape(-1))


def read_bf(path, cap_bits=None):
    with open(path, "rb") as f:
        bits = int(np.fromfile(f, "<u8", 1)[0])
        keep = bits if cap_bits is None else min(bits, cap_bits)
        words = np.fromfile(f, "<u8", (keep + 63) // 64)
    return bits, words


def read_index(prefix):
    a = {}
    a.update(read_ref_dict(prefix + ".ref.dict"))
    a.update(read_snp_dict(prefix + ".snp.dict"))
    a["ref_bf_bits"], a["ref_bf_words"] = read_bf(prefix + ".ref.bf", 1 << 32)
    a["snp_bf_bits"], a["snp_bf_words"] = read_bf(prefix + ".snp.bf")
    return a


def read_chrlens(path):
    out = []
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) >= 2:
                out.append((p[0][:32], int(p[1])))
    return out


def write_bf_sparse(path, bits, set_positions):
    """Write a bit-vector file of `bits` bits with the given bits set, as a sparse file (holes for
    the zero pages), so a 1.2 GB reference-format file costs a few KB of disk in tests."""
    nwords = (int(bits) + 63) // 64
    pos = np.asarray(set_positions, dtype=np.uint64)
    widx = (pos >> np.uint64(6)).astype(np.int64)
    with open(path, "wb") as f:
        f.write(np.uint64(bits).tobytes())
        f.truncate(8 + 8 * nwords)
        if len(pos):
            order = np.argsort(widx, kind="stable")
            widx, pos = widx[order], pos[order]
            uniq, start = np.unique(widx, return_index=True)
            vals = np.bitwise_or.reduceat(np.uint64(1) << (pos & np.uint64(63)), start)
            for w, v in zip(uniq.tolist(), vals.tolist()):
                f.seek(8 + 8 * w)
                f.write(np.uint64(v).tobytes())


def bf_set_positions(words):
    nz = np.nonzero(words)[0]
    out = []
    for i in nz.tolist():
        v = int(words[i])
        while v:
            b = (v & -v).bit_length() - 1
            out.append(i * 64 + b)
            v &= v - 1
    return np.array(out, dtype=np.uint64)
