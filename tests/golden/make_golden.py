#!/usr/bin/env python3
"""Regenerates tests/golden/ from the REAL reference (oracle/_ref/vargeno, built by oracle/Makefile
from /root/reference).  Runs only in the build container; the GPU box and the tests use the
committed outputs.  Usage:  python tests/golden/make_golden.py [workdir]

Produces
  ftiny.*      60 kbp / 2 951 SNPs / 4 000 reads -- committed whole: inputs, the reference-written
               dict files, the set bits of its bit-vector files, its output VCF.
  ftiny.info_<kind>.sha256  the reference's index files when the SNP list's INFO column is one of vcf_variants.INFO_KINDS
               (records without a CAF key after records with one at another place; no CAF on the first record; unknown names)
  ftiny.out.<kind>.vcf.gz   the reference's output when the SNP list it annotates is one of tests/vcf_variants.py
               (GT already declared, FORMAT/sample columns present, "chr"-prefixed names, blank lines)
  fsmall.*     F-small of SURVEY.md §8c (300 kbp / 29 868 SNPs / 40 000 reads): inputs are a pure
               function of the seed (vargeno_amd/synth.py), so only the sha256 list and the
               reference's output VCF are committed.
  fdense.*     dense-bucket fixture (synth.f_dense: ~150 SNP k-mers per HI24 bucket, the bucket shape of hg38 + full dbSNP):
               sha256 list + the reference's output VCF.
  fquirk.*     irregular FASTA / VCF inputs (synth.f_quirk: IUPAC codes, blanks and carriage returns in sequence lines, long
               names, multi-allelic / indel / out-of-order / repeated records, odd INFO fields ...): sha256 list + the
               reference's output VCF.
  fstrands.*   strand corner cases (synth.f_strands: reference and SNP k-mers that are their own reverse complement, k-mers
               whose reverse complement is in the dictionary too, reads of both strands aimed at them): sha256 list + the
               reference's output VCF.
  frepeated.*  synth.f_repeated_records: an SNP list holding the same record one to five times (auxiliary rows that list one position
               several times; a chunk voting several times for a position): sha256 list + the reference's VCF
  flowcomplex.*  synth.f_lowcomplex(1): microsatellites, hairpins, tandem and dispersed copies, dense SNPs: sha256 list + the
               reference's output VCF.
Fixtures are data (inputs and reference outputs); no reference source text is stored here.
"""
import gzip
import hashlib
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vargeno_amd import synth  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "vargeno")
OUT = os.path.dirname(os.path.abspath(__file__))


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def gz(src, dst):
    with open(src, "rb") as f, gzip.GzipFile(dst, "wb", compresslevel=9, mtime=0) as g:
        shutil.copyfileobj(f, g)


def bf_setbits(path):
    w = np.fromfile(path, dtype=np.uint64)
    bits, words = int(w[0]), w[1:]
    nz = np.nonzero(words)[0]
    pos = []
    for i in nz:
        v = int(words[i])
        while v:
            b = (v & -v).bit_length() - 1
            pos.append(int(i) * 64 + b)
            v &= v - 1
    return bits, np.array(pos, dtype=np.uint64)


def run(name, gen, work, commit_all):
    d = os.path.join(work, name)
    os.makedirs(d, exist_ok=True)
    if name in ("fquirk", "frepeated"):
        synth.write_quirk(d, gen())
    else:
        g, s, r = gen()[:3]
        synth.write_fasta(os.path.join(d, "ref.fa"), g)
        synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
        synth.write_fastq(os.path.join(d, "reads.fq"), r)
    subprocess.check_call([REF_BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, stdout=subprocess.DEVNULL)
    subprocess.check_call([REF_BIN, "geno", "idx", "reads.fq", "snps.vcf", "out.vcf"], cwd=d, stdout=subprocess.DEVNULL)
    files = ["ref.fa", "snps.vcf", "reads.fq", "idx.chrlens", "idx.ref.dict", "idx.snp.dict", "idx.ref.bf",
             "idx.snp.bf", "out.vcf"]
    with open(os.path.join(OUT, name + ".sha256"), "w") as f:
        for fn in files:
            f.write("%s  %s\n" % (sha(os.path.join(d, fn)), fn))
    gz(os.path.join(d, "out.vcf"), os.path.join(OUT, name + ".out.vcf.gz"))
    if commit_all:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import vcf_variants

        text = open(os.path.join(d, "snps.vcf")).read()
        for kind in vcf_variants.KINDS:
            with open(os.path.join(d, "snps.%s.vcf" % kind), "w") as f:
                f.write(vcf_variants.make(text, kind))
            subprocess.check_call([REF_BIN, "geno", "idx", "reads.fq", "snps.%s.vcf" % kind, "out.%s.vcf" % kind], cwd=d, stdout=subprocess.DEVNULL)
            gz(os.path.join(d, "out.%s.vcf" % kind), os.path.join(OUT, "%s.out.%s.vcf.gz" % (name, kind)))
        for kind in vcf_variants.INFO_KINDS:                                 # INFO-column variants, index side: sha256 of the reference's files
            with open(os.path.join(d, "info_%s.vcf" % kind), "w") as f:
                f.write(vcf_variants.info_variant(text, kind))
            subprocess.check_call([REF_BIN, "index", "ref.fa", "info_%s.vcf" % kind, "ix_" + kind], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            with open(os.path.join(OUT, "%s.info_%s.sha256" % (name, kind)), "w") as f:
                for ext in ("chrlens", "ref.dict", "snp.dict", "ref.bf", "snp.bf"):
                    f.write("%s  idx.%s\n" % (sha(os.path.join(d, "ix_%s.%s" % (kind, ext))), ext))
        for fn in ("ref.fa", "snps.vcf", "reads.fq", "idx.ref.dict", "idx.snp.dict"):
            gz(os.path.join(d, fn), os.path.join(OUT, "%s.%s.gz" % (name, fn)))
        shutil.copy(os.path.join(d, "idx.chrlens"), os.path.join(OUT, name + ".idx.chrlens"))
        rb, rp = bf_setbits(os.path.join(d, "idx.ref.bf"))
        sb, sp = bf_setbits(os.path.join(d, "idx.snp.bf"))
        np.savez_compressed(os.path.join(OUT, name + ".bf.npz"), ref_bits=np.uint64(rb), ref_set=rp,
                            snp_bits=np.uint64(sb), snp_set=sp)


if __name__ == "__main__":
    work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/vg_golden"
    if not os.path.exists(REF_BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["ftiny", "fsmall", "fdense", "fquirk", "fstrands", "flowcomplex", "frepeated"]
    if "ftiny" in which:
        run("ftiny", synth.f_tiny, work, True)
    if "fsmall" in which:
        run("fsmall", synth.f_small, work, False)
    if "fdense" in which:
        run("fdense", synth.f_dense, work, False)
    if "fquirk" in which:
        run("fquirk", synth.f_quirk, work, False)
    if "fstrands" in which:
        run("fstrands", synth.f_strands, work, False)
    if "flowcomplex" in which:
        run("flowcomplex", lambda: synth.f_lowcomplex(1), work, False)
    if "frepeated" in which:
        run("frepeated", synth.f_repeated_records, work, False)
    # the reference's own test data (test/snp.vcf, test/expected_output): data files, copied verbatim
    if os.path.isdir("/root/reference/test"):
        shutil.copy("/root/reference/test/snp.vcf", os.path.join(OUT, "reftest.snp.vcf"))
        shutil.copy("/root/reference/test/expected_output", os.path.join(OUT, "reftest.expected_output"))
