"""Parity of the HIP path (through the C-ABI) against the oracle and the reference's golden output.
Bit-exact: this is integer / index work, there is no tolerance anywhere."""
import os
import re

import numpy as np
import pytest

from conftest import BIN, GOLDEN
from oracle import oracle as O
from vargeno_amd import index_io
from vargeno_amd.api import GenoIndex

pytestmark = pytest.mark.gpu

CMP_STATS = ["reads", "reads_n", "reads_invalid", "passes", "passes_ok", "chunks", "gate_open", "refbf_pos", "snpbf_pos",
             "large_block", "ref_query", "snp_query", "ref_probe", "snp_probe", "scan_ref", "scan_snp", "scan_oob",
             "aux_ref", "aux_snp", "site_test", "ctx", "walks", "incr", "ingest_bytes"]


def _oracle_counts(prefix, r):
    ox = O.OracleIndex.load(prefix)
    bad = ox.process(r.bases, r.quals, r.offsets)
    return ox, bad, ox.sites()


def test_ftiny_counts_sites_stats_equal_oracle(ftiny_dir, ftiny_reads):
    prefix = os.path.join(ftiny_dir, "idx")
    r = ftiny_reads
    ox, bad, so = _oracle_counts(prefix, r)
    with GenoIndex.open(prefix) as gx:
        sg = gx.sites()
        for k in ("pos", "ref_base", "alt_base", "ref_freq", "alt_freq"):
            assert np.array_equal(sg[k], so[k]), k
        gx.submit(r.bases, r.quals, r.offsets)
        rc, ac = gx.counts()
        assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"])
        st = gx.stats()
        want = ox.stats.as_dict()
        for k in CMP_STATS:
            assert st[k] == want[k], k
        assert st["alg_bytes"] == ox.alg_bytes()
        # and straight against the reference's own VCF
        sg.update(ref_cnt=rc, alt_cnt=ac)
        mine = O.calls_by_key(sg, index_io.read_chrlens(prefix + ".chrlens"))
        assert mine == O.parse_vcf_calls(os.path.join(GOLDEN, "ftiny.out.vcf.gz"))


def test_index_from_memory_equals_index_from_files(ftiny_dir, ftiny_reads):
    """vg_index_create (the dictionaries and bit vectors handed over as arrays, field by field as the files hold them) builds
    the same resident index as vg_index_open (the files' bytes unpacked on the device): same sites, same counters."""
    prefix = os.path.join(ftiny_dir, "idx")
    r = ftiny_reads
    _, _, so = _oracle_counts(prefix, r)
    arrays = index_io.read_index(prefix)
    with GenoIndex.create(arrays) as gm, GenoIndex.open(prefix) as gf:
        sm, sf = gm.sites(), gf.sites()
        for k in ("pos", "ref_base", "alt_base", "ref_freq", "alt_freq"):
            assert np.array_equal(sm[k], sf[k]) and np.array_equal(sm[k], so[k]), k
        assert gm.device_bytes == gf.device_bytes
        gm.submit(r.bases, r.quals, r.offsets)
        rc, ac = gm.counts()
        assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"])


def test_stats_off_build_gives_same_counts(ftiny_dir, ftiny_reads):
    prefix = os.path.join(ftiny_dir, "idx")
    r = ftiny_reads
    _, _, so = _oracle_counts(prefix, r)
    with GenoIndex.open(prefix) as gx:
        gx.set_stats(False)
        gx.submit(r.bases, r.quals, r.offsets)
        rc, ac = gx.counts()
        assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"])
        tm = gx.timing()
        assert tm["ms_main"] > 0 and tm["batches"] == 1


@pytest.mark.parametrize("knob", ["VG_NO_DIRECT", "VG_NO_MX", "VG_NO_MX+VG_NO_HX", "VG_NO_MX+VG_NO_SNP_JG32", "VG_NO_MX+VG_NO_SEC+VG_NO_PROBE_VIEW", "VG_NO_SEC", "VG_PACK_OVERLAP",
                                  "VG_NO_INGEST_STREAM", "VG_NO_PROBE_VIEW", "VG_NO_SIG_VIEW", "VG_NO_BF_FROM_SEC",
                                  # r06: tables that scale with the index.  A small fixture gets small tables by itself (2^22 buckets here); these force the
                                  # 2^32-entry forms an hg38-scale index gets (the headline kernel instantiation), a direct table of 2^16 buckets (~35 entries
                                  # each: the in-bucket bisection) and a 2^16-entry reference jump table (long coarse buckets: ref_bounds' bisection)
                                  "VG_NO_SSEC",      # r06: without the SNP dictionary's LO32-ordered view (the high-half SNP queries one by one, as through r05)
                                  "VG_DX_BITS=32+VG_REF_JG_BITS=32", "VG_DX_BITS=16", "VG_REF_JG_BITS=16", "VG_DX_BITS=19+VG_NO_SEC", "VG_REF_JG_BITS=16+VG_NO_MX+VG_NO_HX"])
def test_fallback_layouts_give_same_counts(ftiny_dir, ftiny_reads, monkeypatch, knob):
    """The timed kernel reads re-laid-out views of the dictionaries (direct table, merged view, LO32-ordered view, strided-probe
    view); the pack kernel can run on the ingest stream.  Each has a fallback / alternative; all must give the reference's bits.
    Several batches, so that slots, streams and the base-indexed counters' fold are exercised too."""
    for k in knob.split("+"):                      # (VG_NO_MX: the kernel of an index too big for the merged view; with
        monkeypatch.setenv(*(k.split("=") if "=" in k else (k, "1")))      # VG_NO_SNP_JG32 it bisects HI24 buckets of the SNP dictionary)
    prefix = os.path.join(ftiny_dir, "idx")
    r = ftiny_reads
    _, _, so = _oracle_counts(prefix, r)
    with GenoIndex.open(prefix) as gx:
        if "VG_DX_BITS" in knob:
            assert "direct table of 2^%s buckets" % knob.split("VG_DX_BITS=")[1][:2] in gx.plan, gx.plan
        if "VG_REF_JG_BITS" in knob and "VG_NO_MX" not in knob:
            assert "reference jump table: 2^%s entries" % knob.split("VG_REF_JG_BITS=")[1][:2] in gx.plan, gx.plan
        # (F-tiny's FASTA is upper case: its reference bit vector is the LO32 set of its dictionary, which the loader verifies)
        assert ("sec_is_bf" in gx.views) == ("VG_NO_BF_FROM_SEC" not in knob and "VG_NO_SEC" not in knob), gx.views
        assert ("ssec" in gx.views) == ("VG_NO_SSEC" not in knob), gx.views
        gx.set_stats(False)
        step = r.n // 5 + 1
        for lo in range(0, r.n, step):
            sub = r.slice(lo, min(r.n, lo + step))
            gx.submit(sub.bases, sub.quals, sub.offsets)
        rc, ac = gx.counts()
        assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"])
        if "BITS" in knob:                         # (the counting build and the lane machine walk the reference's jump table: its coarse form too)
            gx.reset()
            gx.set_stats(True)
            gx.submit(r.bases, r.quals, r.offsets)
            rc, ac = gx.counts()
            assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"])


def test_batching_and_reset_invariance(ftiny_dir, ftiny_reads):
    prefix = os.path.join(ftiny_dir, "idx")
    r = ftiny_reads
    _, _, so = _oracle_counts(prefix, r)
    with GenoIndex.open(prefix) as gx:
        cuts = [0, 1, 2, 700, 701, 2500, r.n]
        for lo, hi in reversed(list(zip(cuts[:-1], cuts[1:]))):
            s = r.slice(lo, hi)
            gx.submit(s.bases, s.quals, s.offsets)
        rc, ac = gx.counts()
        assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"])
        gx.reset()
        rc, ac = gx.counts()
        assert not rc.any() and not ac.any()
        ox2 = O.OracleIndex.load(prefix)
        for _ in range(7):                               # 7x coverage: exact sums grow, the clamp holds at 63
            gx.submit(r.bases, r.quals, r.offsets)
            ox2.process(r.bases, r.quals, r.offsets)
        rc2, ac2 = gx.counts()
        s2 = ox2.sites()
        assert np.array_equal(rc2, s2["ref_cnt"]) and np.array_equal(ac2, s2["alt_cnt"])
        assert rc2.max() == 63


def test_many_ragged_batches_both_builds(ftiny_dir, ftiny_reads):
    """Hundreds of batches of 1..400 reads (slots and streams rotate, most launches are smaller than one workgroup's work
    pool, the base-indexed counters are folded once at the end): counters and event counts equal the oracle's, for the
    counting build and for the timed build."""
    prefix = os.path.join(ftiny_dir, "idx")
    r = ftiny_reads
    ox, _, so = _oracle_counts(prefix, r)
    rng = np.random.default_rng(11)
    cuts = [0]
    while cuts[-1] < r.n:
        cuts.append(min(r.n, cuts[-1] + int(rng.integers(1, 401))))
    want = ox.stats.as_dict()
    for stats in (True, False):
        with GenoIndex.open(prefix) as gx:
            gx.set_stats(stats)
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                s = r.slice(lo, hi)
                gx.submit(s.bases, s.quals, s.offsets)
            rc, ac = gx.counts()
            assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"]), "stats=%s" % stats
            if stats:
                st = gx.stats()
                for k in CMP_STATS:
                    assert st[k] == want[k], k


def test_scratch_overflow_path_is_exact(ftiny_dir, ftiny_reads, monkeypatch):
    """Generic lane tier alone, with a tiny first scratch: lanes that run out hand the read to the
    deep-scratch launch (the path the wave tiers fall back to)."""
    prefix = os.path.join(ftiny_dir, "idx")
    r = ftiny_reads
    ox, _, so = _oracle_counts(prefix, r)
    monkeypatch.setenv("VG_FORCE_GENERIC", "1")
    monkeypatch.setenv("VG_SCRATCH_CAP", "3")
    monkeypatch.setenv("VG_SCRATCH_KCAP", "2")
    with GenoIndex.open(prefix) as gx:
        gx.submit(r.bases, r.quals, r.offsets)
        rc, ac = gx.counts()
        st = gx.stats()
        assert st["overflow_reads"] == 0 and st["overflow_deep"] > 0
        assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"])
        want = ox.stats.as_dict()
        for k in CMP_STATS:
            assert st[k] == want[k], k


def test_generic_lane_tier_alone_is_exact(ftiny_dir, ftiny_reads, monkeypatch):
    """VG_FORCE_GENERIC=1 bypasses the wave-cooperative kernel: both tiers must give the same bits."""
    prefix = os.path.join(ftiny_dir, "idx")
    r = ftiny_reads
    ox, _, so = _oracle_counts(prefix, r)
    monkeypatch.setenv("VG_FORCE_GENERIC", "1")
    with GenoIndex.open(prefix) as gx:
        gx.submit(r.bases, r.quals, r.offsets)
        rc, ac = gx.counts()
        st = gx.stats()
        assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"])
        want = ox.stats.as_dict()
        for k in CMP_STATS:
            assert st[k] == want[k], k
        assert st["overflow_reads"] == 0


def test_wave_tier_takes_most_reads(ftiny_dir, ftiny_reads):
    prefix = os.path.join(ftiny_dir, "idx")
    r = ftiny_reads
    with GenoIndex.open(prefix) as gx:
        gx.submit(r.bases, r.quals, r.offsets)
        st = gx.stats()
        assert 0 < st["overflow_reads"] < r.n // 2          # the adversarial fixture does spill some reads to the deep-list tier
        assert st["overflow_deep"] <= st["overflow_reads"]


def test_fsmall_full_parity_through_all_tiers(tmp_path):
    """F-small (40 % gate-open chunks, planted repeats, 400-entry SNP buckets, >10-copy k-mers): index built
    by the product, counters and event counts equal to the oracle, and every tier sees work."""
    import subprocess
    from conftest import ROOT
    from vargeno_amd import synth
    g, s, r = synth.f_small()
    d = str(tmp_path)
    synth.write_fasta(os.path.join(d, "ref.fa"), g)
    synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d,
                          env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
    prefix = os.path.join(d, "idx")
    ox, _, so = _oracle_counts(prefix, r)
    with GenoIndex.open(prefix) as gx:
        gx.submit(r.bases, r.quals, r.offsets)
        rc, ac = gx.counts()
        st = gx.stats()
        assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"])
        want = ox.stats.as_dict()
        for k in CMP_STATS:
            assert st[k] == want[k], k
        assert st["overflow_reads"] > 0 and st["large_block"] > 0 and st["scan_oob"] > 0
        print("fsmall tiers: spilled %d of %d reads, %d to the lane tier" % (st["overflow_reads"], r.n, st["overflow_deep"]))
        # the same batch with its quality strings reduced to one gate word per read (vg_reads_process_device_gated): reads of
        # 31-250 bases, 40 % gate-open chunks, every tier -- both builds of the kernel
        import torch
        from vargeno_amd.api import gate_words

        dev = torch.device("cuda", 0)
        tb, tq = torch.from_numpy(r.bases).to(dev), torch.from_numpy(r.quals).to(dev)
        to = torch.from_numpy(r.offsets.astype(np.int64)).to(dev)
        gw = gate_words(tq, to)
        lens = (r.offsets[1:] - r.offsets[:-1]).astype(np.int64)
        o = r.offsets.astype(np.int64)
        expect = np.zeros(r.n, np.int64)
        for c in range(int(lens.max()) // 32):
            live = lens // 32 > c
            expect[live] |= (r.quals[o[:-1][live] + c] < ord("8")).astype(np.int64) << c
        assert np.array_equal(gw.cpu().numpy().view(np.uint32).astype(np.int64), expect)
        del tq
        for stats in (True, False):
            gx.reset()
            gx.set_stats(stats)
            gx.process_device_gated(tb, gw, to, r.n)
            rc, ac = gx.counts()
            assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"]), "gate words, stats=%s" % stats
            if stats:
                st = gx.stats()
                for k in CMP_STATS:
                    assert st[k] == want[k], k


def test_dense_snp_buckets_parity_all_layouts(tmp_path, monkeypatch):
    """synth.f_dense: ~150 SNP-dictionary entries per HI24 bucket (the bucket shape of BASELINE.json configs[4], hg38 + full
    dbSNP) and dense HI32 / LO32 buckets -- stage B's strided scans run hundreds of probes per gate-open chunk, merged-view
    buckets need the bisection path, LO32-view runs exceed what one lane walks.  Counters and event counts equal the
    oracle's (itself pinned on this fixture against the reference, tests/test_oracle_golden.py) for the counting build, the
    timed build, and the layout a > 2^32-entry index falls back to (no merged view / direct table)."""
    import subprocess
    import time

    from vargeno_amd import synth

    g, s, r = synth.f_dense()
    d = str(tmp_path)
    synth.write_fasta(os.path.join(d, "ref.fa"), g)
    synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
    prefix = os.path.join(d, "idx")
    ox = O.OracleIndex.load(prefix)
    ox.process(r.bases, r.quals, r.offsets, nthreads=8)
    so = ox.sites()
    want = ox.stats.as_dict()
    assert want["scan_snp"] > 100 * want["gate_open"]
    rows = []
    for label, env, views in (("all views", {}, ("mx", "dx")), ("all views, tables of 2^32 entries (the hg38-scale forms)", {"VG_DX_BITS": "32", "VG_REF_JG_BITS": "32"}, ("mx", "dx")),
                              ("all views, direct table of 2^16 buckets", {"VG_DX_BITS": "16", "VG_REF_JG_BITS": "16"}, ("mx", "dx")),
                              ("no merged view (the > 2^32-entry fallback): paired HI32 table", {"VG_NO_MX": "1"}, ("hx",)),
                              ("no merged view, HI32 jump tables instead of the paired table", {"VG_NO_MX": "1", "VG_NO_HX": "1"}, ("snp_jg32",)),
                              ("no merged view, no HI32 table of the SNP dictionary at all", {"VG_NO_MX": "1", "VG_NO_HX": "1", "VG_NO_SNP_JG32": "1"}, ())):
        for k in ("VG_DX_BITS", "VG_REF_JG_BITS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with GenoIndex.open(prefix) as gx:
            assert all(v in gx.views for v in views) and ("mx" in gx.views) == ("mx" in views), (label, gx.views)
            gx.submit(r.bases, r.quals, r.offsets)                   # counting build
            rc, ac = gx.counts()
            st = gx.stats()
            assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"]), label
            for k in CMP_STATS:
                assert st[k] == want[k], (label, k)
            gx.reset()
            gx.set_stats(False)                                      # timed build
            t0 = time.time()
            gx.submit(r.bases, r.quals, r.offsets)
            rc, ac = gx.counts()
            rows.append((label, time.time() - t0, gx.timing()["ms_main"], st["overflow_reads"]))
            assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"]), label
    for row in rows:
        print("f_dense %-45s %.3f s wall, wave kernel %.3f ms, %d of %d reads spilled" % (row + (r.n,)))


def test_edge_reads(ftiny_dir):
    prefix = os.path.join(ftiny_dir, "idx")
    reads = [b"", b"ACGT", b"A" * 31, b"ACGTN" * 10, b"ACGT" * 8 + b"N", b"ACGX" * 8, b"acgt" * 16,
             b"ACGTACGTACGTACGTACGTACGTACGTACGN" + b"X" * 32, b"T" * 1022]
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8)
    offs = np.cumsum([0] + [len(x) for x in reads]).astype(np.uint64)
    quals = np.full(len(bases), ord("#"), np.uint8)
    ox = O.OracleIndex.load(prefix)
    ox.process(bases, quals, offs)
    with GenoIndex.open(prefix) as gx:
        gx.submit(bases, quals, offs)
        st = gx.stats()
        want = ox.stats.as_dict()
        for k in CMP_STATS:
            assert st[k] == want[k], k
        assert st["reads_invalid"] == 1 and st["reads_n"] == 2
        so = ox.sites()
        rc, ac = gx.counts()
        assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"])
        # zero reads is a no-op, not an error
        gx.submit(bases[:0], quals[:0], np.zeros(1, np.uint64))


def test_errors_are_codes_not_aborts(ftiny_dir, tmp_path):
    from vargeno_amd._lib import VgError

    with pytest.raises(VgError) as e:
        GenoIndex.open(str(tmp_path / "nope"))
    assert e.value.code == -2
    prefix = os.path.join(ftiny_dir, "idx")
    with GenoIndex.open(prefix) as gx:
        long = np.full(1023, ord("A"), np.uint8)
        with pytest.raises(VgError) as e:
            gx.submit(long, long, np.array([0, 1023], np.uint64))
        assert e.value.code == -6


def test_device_resident_batches_and_very_long_reads(ftiny_dir, ftiny_reads):
    """vg_reads_process_device (what bench.py times) on torch buffers, including reads far beyond the reference's
    1022-base line buffer: 32-chunk reads stay in the wave tier, longer ones take the generic lane tier."""
    import torch
    from vargeno_amd import synth

    prefix = os.path.join(ftiny_dir, "idx")
    g = synth.f_tiny()[0]
    cat = np.concatenate(g.seqs)
    rng = np.random.default_rng(5)
    reads = []
    for L in (1024, 1056, 2000, 5000, 1023, 33):
        st = int(rng.integers(1000, len(cat) - L - 1000))
        reads.append(cat[st:st + L])
    bases = np.concatenate([ftiny_reads.bases] + reads)
    extra_q = rng.integers(ord("#"), ord("I") + 1, size=sum(len(x) for x in reads), dtype=np.uint8)
    quals = np.concatenate([ftiny_reads.quals, extra_q])
    offs = np.concatenate([ftiny_reads.offsets, ftiny_reads.offsets[-1] + np.cumsum([len(x) for x in reads]).astype(np.uint64)])
    ox = O.OracleIndex.load(prefix)
    ox.process(bases, quals, offs)
    so = ox.sites()
    with GenoIndex.open(prefix) as gx:
        dev = torch.device("cuda", 0)
        tb, tq = torch.from_numpy(bases).to(dev), torch.from_numpy(quals).to(dev)
        to = torch.from_numpy(offs.astype(np.int64)).to(dev)
        for stats in (True, False):
            gx.reset()
            gx.set_stats(stats)
            gx.process_device(tb, tq, to, len(offs) - 1)
            rc, ac = gx.counts()
            assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"]), stats
        gx.set_stats(True)
        gx.reset()
        gx.process_device(tb, tq, to, len(offs) - 1)
        st = gx.stats()
        want = ox.stats.as_dict()
        for k in CMP_STATS:
            assert st[k] == want[k], k


@pytest.mark.parametrize("seed", [11, 12])
def test_randomly_damaged_ragged_reads_both_builds(ftiny_dir, seed):
    """Seeded fuzz: reads of every length from 0 to 400 cut from the genome (both strands), with substitutions, lower case, N
    runs, the odd non-ACGTN character, and quality lines from all-low to all-high, in batches of odd sizes -- every event
    counter and every site counter of both builds against the oracle."""
    from vargeno_amd import synth

    prefix = os.path.join(ftiny_dir, "idx")
    g = synth.f_tiny()[0]
    cat = np.concatenate(g.seqs)
    rng = np.random.default_rng(seed)
    comp = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
        comp[a] = b
    reads, quals = [], []
    for _ in range(6000):
        L = int(rng.integers(0, 401)) if rng.random() < 0.5 else int(rng.choice([31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 150, 151, 160, 161, 250]))
        st = int(rng.integers(0, len(cat) - L))
        s = cat[st:st + L].copy()
        if rng.random() < 0.5:
            s = comp[s[::-1]]
        k = rng.random()
        if L and k < 0.6:                                       # a few substitutions
            for p in rng.integers(0, L, size=int(rng.integers(0, 4))):
                s[p] = rng.choice(np.frombuffer(b"ACGT", np.uint8))
        if L and rng.random() < 0.15:
            s[rng.random(L) < 0.3] |= 0x20                      # lower case is accepted (util.c:89-111)
        if L and rng.random() < 0.05:
            p = int(rng.integers(0, L)); s[p:p + int(rng.integers(1, 5))] = ord("N")
        if L and rng.random() < 0.01:
            s[int(rng.integers(0, L))] = rng.choice(np.frombuffer(b"XR-.*", np.uint8))
        lowq = rng.choice([0.0, 0.08, 0.5, 1.0])
        q = np.where(rng.random(L) < lowq, rng.integers(ord("#"), ord("8"), size=L), rng.integers(ord("8"), ord("J"), size=L)).astype(np.uint8)
        reads.append(s); quals.append(q)
    bases = np.concatenate(reads); qs = np.concatenate(quals)
    offs = np.concatenate([[0], np.cumsum([len(x) for x in reads])]).astype(np.uint64)
    keep = np.diff(offs.astype(np.int64)) <= 1022              # the reference's line buffer (qv.cc:700); longer reads are refused at submit
    assert keep.all()
    ox = O.OracleIndex.load(prefix)
    ox.process(bases, qs, offs)
    so, want = ox.sites(), ox.stats.as_dict()
    assert want["reads_invalid"] > 0 and want["reads_n"] > 0 and want["incr"] > 0
    cuts = [0, 1, 2, 65, 66, 1000, 1003, 4097, len(reads)]
    with GenoIndex.open(prefix) as gx:
        for stats in (True, False):
            gx.reset()
            gx.set_stats(stats)
            for a, b in zip(cuts[:-1], cuts[1:]):
                lo, hi = int(offs[a]), int(offs[b])
                gx.submit(bases[lo:hi], qs[lo:hi], offs[a:b + 1] - offs[a])
            rc, ac = gx.counts()
            assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"]), "stats=%s" % stats
            st = gx.stats()
            if stats:
                for k in CMP_STATS:
                    assert st[k] == want[k], k
            assert st["reads_invalid"] == want["reads_invalid"]


def test_corrupt_dictionaries_are_refused_not_read(ftiny_dir, tmp_path):
    """An index whose files have the right sizes but the wrong contents -- k-mers out of order, an entry naming an auxiliary row
    beyond the table -- must come back as an error code from vg_index_open, not as a wild read on the device."""
    import shutil

    from vargeno_amd._lib import VgError

    def copy_index(dst):
        os.makedirs(dst, exist_ok=True)
        for fn in ("idx.ref.dict", "idx.snp.dict", "idx.ref.bf", "idx.snp.bf", "idx.chrlens"):
            shutil.copy(os.path.join(ftiny_dir, fn), os.path.join(dst, fn))
        return os.path.join(dst, "idx")

    # reference dictionary: u64 n, u64 n_aux, then n x {u64 kmer, u32 pos, u8 ambig} (13 bytes)
    p = copy_index(str(tmp_path / "swapped"))
    raw = bytearray(open(p + ".ref.dict", "rb").read())
    a, b = 16 + 13 * 100, 16 + 13 * 101
    raw[a:a + 8], raw[b:b + 8] = raw[b:b + 8], raw[a:a + 8]
    open(p + ".ref.dict", "wb").write(raw)
    with pytest.raises(VgError) as e:
        GenoIndex.open(p)
    assert e.value.code == -2 and "out of order" in str(e.value)
    p = copy_index(str(tmp_path / "wild"))
    raw = bytearray(open(p + ".ref.dict", "rb").read())
    n = int.from_bytes(raw[0:8], "little")
    k = next(i for i in range(n) if raw[16 + 13 * i + 12] != 0 and raw[16 + 13 * i + 8:16 + 13 * i + 12] != b"\xff\xff\xff\xff")
    raw[16 + 13 * k + 8:16 + 13 * k + 12] = (0x7FFFFFF0).to_bytes(4, "little")
    open(p + ".ref.dict", "wb").write(raw)
    with pytest.raises(VgError) as e:
        GenoIndex.open(p)
    assert e.value.code == -2 and "auxiliary rows" in str(e.value)


def _revcomp_keys(k):
    """Reverse complement of 32-mers packed two bits a base (A C G T = 0 1 2 3): complement, then reverse the 2-bit fields."""
    k = ~k.astype(np.uint64)
    for sh, mask in ((2, 0x3333333333333333), (4, 0x0F0F0F0F0F0F0F0F), (8, 0x00FF00FF00FF00FF), (16, 0x0000FFFF0000FFFF)):
        m = np.uint64(mask)
        k = ((k >> np.uint64(sh)) & m) | ((k & m) << np.uint64(sh))
    return (k >> np.uint64(32)) | (k << np.uint64(32))


@pytest.mark.parametrize("knob", [None, "VG_NO_DIRECT", "VG_NO_MX", "VG_DX_BITS=32"])
def test_self_complementary_and_both_strand_kmers(tmp_path, monkeypatch, knob):
    """The merged view and the direct table are keyed by min(K, revcomp K) with a strand flag, so that one look-up answers both
    passes of a read.  The corner cases of that: 32-mers that are their own reverse complement (an entry of BOTH strands), at
    one position and at two, 32-mers whose reverse complement is in the dictionary too (one key, entries of either strand),
    reverse-strand reads that run pass 1 on pass 0's look-ups.  synth.f_strands plants them and aims reads of both strands
    at them, exact and one substitution away; counters and event counts must equal the oracle's (which knows nothing of
    canonical keys: it follows qv.cc:760-1558 with the reference's two dictionaries) in the counting and the timed build,
    with the direct table, with the merged view alone and with neither."""
    import subprocess

    from vargeno_amd import synth

    g, s, r, plants = synth.f_strands()
    d = str(tmp_path)
    synth.write_fasta(os.path.join(d, "ref.fa"), g)
    synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
    prefix = os.path.join(d, "idx")
    rk = index_io.read_ref_dict(prefix + ".ref.dict")["ref_kmer"]
    sk = index_io.read_snp_dict(prefix + ".snp.dict")["snp_kmer"]
    n_self = int((_revcomp_keys(rk) == rk).sum())
    n_both = int(np.isin(_revcomp_keys(rk), rk).sum()) - n_self
    assert n_self >= len(plants) // 3 and n_both >= len(plants) // 2, (n_self, n_both)
    ox = O.OracleIndex.load(prefix)
    ox.process(r.bases, r.quals, r.offsets, nthreads=8)
    so = ox.sites()
    want = ox.stats.as_dict()
    assert so["ref_cnt"].sum() + so["alt_cnt"].sum() > 10_000 and want["snp_probe"] > 0
    if knob:
        monkeypatch.setenv(*(knob.split("=") if "=" in knob else (knob, "1")))
    with GenoIndex.open(prefix) as gx:
        assert ("dx" in gx.views) == (knob is None or "BITS" in knob) and ("mx" in gx.views) == (knob != "VG_NO_MX"), gx.views
        for stats in (True, False):
            gx.reset()
            gx.set_stats(stats)
            gx.submit(r.bases, r.quals, r.offsets)
            rc, ac = gx.counts()
            bad = np.nonzero((rc != so["ref_cnt"]) | (ac != so["alt_cnt"]))[0]
            assert len(bad) == 0, ("stats=%s" % stats, so["pos"][bad[:10]], rc[bad[:10]], so["ref_cnt"][bad[:10]], ac[bad[:10]], so["alt_cnt"][bad[:10]])
            if stats:
                st = gx.stats()
                for k in CMP_STATS:
                    assert st[k] == want[k], k
    for fn in ("idx.ref.bf", "idx.snp.bf"):
        os.remove(os.path.join(d, fn))


@pytest.mark.parametrize("seed,knob", [(1, None), (2, None), (3, None), (4, "VG_NO_MX"), (5, "VG_NO_DIRECT"), (6, "VG_NO_MX+VG_NO_HX"), (7, "VG_FORCE_AUX_DUPS")])
def test_low_complexity_genomes(tmp_path, monkeypatch, seed, knob):
    """synth.f_lowcomplex: genomes made of microsatellites (runs of A, AT, ACGT are their own reverse complement), hairpins,
    tandem and dispersed copies (auxiliary rows, POS_AMBIGUOUS), one SNP per 25 bases -- k-mers that collide with themselves,
    with their reverse complements and with each other, a different mix for every seed.  (On seeds 1-15 the oracle's calls
    were checked against the reference binary's in the build container; tests/test_oracle_golden.py keeps seed 1.)  Counters
    and event counts of both builds equal the oracle's, on the shipped layout and on the fall-back ones."""
    import subprocess

    from vargeno_amd import synth

    g, s, r = synth.f_lowcomplex(seed)
    d = str(tmp_path)
    synth.write_fasta(os.path.join(d, "ref.fa"), g)
    synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
    prefix = os.path.join(d, "idx")
    ox = O.OracleIndex.load(prefix)
    ox.process(r.bases, r.quals, r.offsets, nthreads=8)
    so = ox.sites()
    want = ox.stats.as_dict()
    assert want["aux_ref"] > 500 and so["ref_cnt"].sum() + so["alt_cnt"].sum() > 5_000
    for k in (knob.split("+") if knob else ()):
        monkeypatch.setenv(k, "1")
    with GenoIndex.open(prefix) as gx:
        for stats in (True, False):
            gx.reset()
            gx.set_stats(stats)
            gx.submit(r.bases, r.quals, r.offsets)
            rc, ac = gx.counts()
            bad = np.nonzero((rc != so["ref_cnt"]) | (ac != so["alt_cnt"]))[0]
            assert len(bad) == 0, ("stats=%s" % stats, so["pos"][bad[:10]], rc[bad[:10]], so["ref_cnt"][bad[:10]], ac[bad[:10]], so["alt_cnt"][bad[:10]])
            if stats:
                st = gx.stats()
                for k in CMP_STATS:
                    assert st[k] == want[k], k
    for fn in ("idx.ref.bf", "idx.snp.bf"):
        os.remove(os.path.join(d, fn))


@pytest.mark.parametrize("knob", [None, "VG_NO_DIRECT", "VG_NO_MX"])
def test_snp_records_held_several_times(tmp_path, monkeypatch, knob):
    """synth.f_repeated_records: the same SNP record one to five times in the list -> auxiliary rows that list ONE position
    several times -> a chunk that votes for, and walks, the same context several times (qv.cc:913-933, 1444-1494).  The wave
    kernel's key table gives a chunk one vote per key: the loader finds such rows (DevIndex::aux_dups, named in vg_index_plan),
    rows are then expanded column by column and the reads concerned end in the lane machine, which keeps contexts one by one.
    Counters and event counts of both builds equal the oracle's (which equals the reference's VCF on this fixture:
    tests/test_oracle_golden.py); the CLI's VCF is compared with the reference's bytes in tests/test_gpu_cli.py."""
    import subprocess

    from vargeno_amd import synth

    q = synth.f_repeated_records()
    d = str(tmp_path)
    synth.write_quirk(d, q)
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
    prefix = os.path.join(d, "idx")
    r = q["reads"]
    ox = O.OracleIndex.load(prefix)
    ox.process(r.bases, r.quals, r.offsets, nthreads=8)
    so = ox.sites()
    want = ox.stats.as_dict()
    assert want["aux_snp"] > 500 and so["ref_cnt"].sum() + so["alt_cnt"].sum() > 1_000
    if knob:
        monkeypatch.setenv(knob, "1")
    with GenoIndex.open(prefix) as gx:
        assert "repeat a position" in gx.plan, gx.plan
        for stats in (True, False):
            gx.reset()
            gx.set_stats(stats)
            gx.submit(r.bases, r.quals, r.offsets)
            rc, ac = gx.counts()
            bad = np.nonzero((rc != so["ref_cnt"]) | (ac != so["alt_cnt"]))[0]
            assert len(bad) == 0, ("stats=%s" % stats, so["pos"][bad[:10]], rc[bad[:10]], so["ref_cnt"][bad[:10]], ac[bad[:10]], so["alt_cnt"][bad[:10]])
            if stats:
                st = gx.stats()
                for k in CMP_STATS:
                    assert st[k] == want[k], k
    for fn in ("idx.ref.bf", "idx.snp.bf"):
        os.remove(os.path.join(d, fn))


def test_a_device_memory_budget_decides_the_views_and_nothing_else_does(ftiny_dir, ftiny_reads):
    """vg_index_open_ex: the views a replica holds follow from the index and the byte budget alone -- the same budget gives the
    same views every time, whatever else lives on the device (here: a 6 GiB tensor of somebody else's); a smaller budget drops
    views in the documented order (direct table first), says so in vg_index_plan, and the results never change; a budget below
    the smallest layout is refused."""
    import torch

    from vargeno_amd._lib import VgError

    prefix = os.path.join(ftiny_dir, "idx")
    r = ftiny_reads
    _, _, so = _oracle_counts(prefix, r)
    GiB = 1 << 30
    seen = {}
    # r06: the tables' widths are a planning decision -- the widest the budget holds (sparser buckets are faster), down to a quarter of
    # the buckets the index wants.  F-tiny on the whole device gets the 2^32-entry forms of an hg38-scale index (~90 GB); under a
    # budget of a few GiB the same files give a handle of that size, tables of ~2 entries' worth per k-mer, and the same counters
    n_ref = len(index_io.read_ref_dict(prefix + ".ref.dict")["ref_kmer"])
    n_snp = len(index_io.read_snp_dict(prefix + ".snp.dict")["snp_kmer"])
    nat = max(16, int(np.ceil(np.log2(n_ref + n_snp))))                 # (plan_views' table_bits_for)
    with GenoIndex.open(prefix) as gx:
        assert "dx" in gx.views and "nothing left out" in gx.plan and "direct table of 2^32 buckets" in gx.plan and "reference jump table: 2^32 entries" in gx.plan, gx.plan
    last_bits = 33
    for budget in (40 * GiB, 12 * GiB, 5 * GiB, int(3.9 * GiB)):
        with GenoIndex.open(prefix, max_device_bytes=budget) as gx:
            m = re.search(r"direct table of 2\^(\d+) buckets", gx.plan)
            assert "dx" in gx.views and m, gx.plan
            b_ = int(m.group(1))
            assert nat - 4 <= b_ <= min(last_bits, nat + 2) and gx.device_bytes <= budget, (budget, gx.plan, gx.device_bytes)
            last_bits = b_
            gx.set_stats(False)
            gx.submit(r.bases, r.quals, r.offsets)
            rc, ac = gx.counts()
            assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"]), budget
            print("budget %.1f GiB: %.2f GB held | %s" % (budget / GiB, gx.device_bytes / 1e9, gx.plan))
    assert last_bits <= nat + 2
    # the rest of the test: the 2^32-entry forms of an hg38-scale index, forced on the small fixture (the budgets are theirs)
    import pytest as _pt
    mp = _pt.MonkeyPatch()
    mp.setenv("VG_DX_BITS", "32")
    mp.setenv("VG_REF_JG_BITS", "32")
    # F-tiny: the smallest layout is planned at ~20.5 GiB (the 16 GiB jump table is always there, 1.8 GiB of lane-tier scratch, 2 GiB
    # reserved for batch slots); the LO32 and signature views are tiny here, the merged view adds 16 GiB, the direct table 48 more
    for budget in (200 * GiB, 60 * GiB, 40 * GiB, 21 * GiB, 200 * GiB, 60 * GiB):
        hog = torch.empty(6 * GiB, dtype=torch.uint8, device="cuda:0") if len(seen) % 2 else None
        with GenoIndex.open(prefix, max_device_bytes=budget) as gx:
            views, plan = tuple(gx.views), gx.plan
            gx.set_stats(False)
            gx.submit(r.bases, r.quals, r.offsets)
            rc, ac = gx.counts()
            assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"]), (budget, views)
            assert gx.device_bytes <= budget
        del hog
        if budget in seen:
            assert seen[budget] == (views, plan), "the same budget gave different views"
        seen[budget] = (views, plan)
        print("budget %3d GiB: %s | %s" % (budget // GiB, ",".join(views), plan))
    assert "dx" in seen[200 * GiB][0] and "mx" in seen[200 * GiB][0] and "nothing left out" in seen[200 * GiB][1]
    assert "dx" not in seen[60 * GiB][0] and "mx" in seen[60 * GiB][0] and "LEFT OUT" in seen[60 * GiB][1] and "direct table" in seen[60 * GiB][1]
    assert "mx" not in seen[21 * GiB][0] and "sec" in seen[21 * GiB][0]
    with pytest.raises(VgError) as e:
        GenoIndex.open(prefix, max_device_bytes=10 * GiB)
    assert e.value.code == -3 and "budget" in str(e.value)
    mp.undo()


@pytest.mark.parametrize("knob", [{}, {"VG_LATE_READS": "2"}, {"VG_NO_LATE_STORE": "1"}])
def test_reads_the_deep_tier_leaves_behind_are_finished_late_or_per_batch(tmp_path, monkeypatch, knob):
    """The few reads that outgrow the deep tier's LDS tables are finished by the lane machine (lists in HBM).  Round 5 ran it per
    batch -- the batch's slot waited 6-12 ms for a handful of 250 bp reads -- round 6 copies their packed form into a store of the
    handle (vg_late_collect) and runs the lane machine over the store once, at the next synchronisation.  synth.f_manykeys sends
    48 reads all the way down; here in three batches and with a synchronisation in the middle: the store as shipped, a store of two reads (it
    fills up: the rest takes the per-batch lane launch, as do reads of more than 32 chunks), and no store at all -- counting build
    and timed build, the oracle's counters and event counts every time."""
    import subprocess
    from vargeno_amd import synth
    g, s, r = synth.f_manykeys()          # (reads with 64 vote keys: ten copies of each of their seven chunks' k-mers -- the deep tier holds 48)
    d = str(tmp_path)
    synth.write_fasta(os.path.join(d, "ref.fa"), g)
    synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
    prefix = os.path.join(d, "idx")
    ox, _, so = _oracle_counts(prefix, r)
    want = ox.stats.as_dict()
    for k, v in knob.items():
        monkeypatch.setenv(k, v)
    parts = [(r.n - 20, r.n), (0, r.n - 48), (r.n - 48, r.n - 20)]       # (the 48 many-key reads are the fixture's last: 20 before the synchronisation, 28 after)
    with GenoIndex.open(prefix) as gx:
        for stats in (True, False):
            gx.reset()
            gx.set_stats(stats)
            for i in range(3):
                part = r.slice(*parts[i])
                gx.submit(part.bases, part.quals, part.offsets)
                if i == 0:
                    gx.sync()                                        # (a run of the store in the middle of the job)
            rc, ac = gx.counts()
            assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"]), (knob, stats)
            st = gx.stats()
            assert st["overflow_deep"] > 2, st["overflow_deep"]      # (the fixture does reach the lane tier, with more reads than the two-read store holds)
            if stats:
                for k in CMP_STATS:
                    assert st[k] == want[k], (knob, k)
