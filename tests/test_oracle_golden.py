"""Pins the oracle (oracle/vg_oracle.c) against outputs of the REAL reference binary captured by
tests/golden/make_golden.py: if these fail, nothing else in the suite means anything."""
import hashlib
import os

import numpy as np

from conftest import GOLDEN, read_sha256_list
from oracle import oracle as O
from vargeno_amd import index_io, synth


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def test_synth_inputs_match_committed_hashes(ftiny_dir, tmp_path):
    """The generator is a pure function of its seed: regenerated inputs == the bytes the reference saw."""
    want = read_sha256_list("ftiny")
    for fn in ("ref.fa", "snps.vcf", "reads.fq", "idx.ref.dict", "idx.snp.dict"):
        assert _sha(os.path.join(ftiny_dir, fn)) == want[fn], fn
    g, s, r = synth.f_tiny()
    synth.write_fasta(str(tmp_path / "ref.fa"), g)
    synth.write_vcf(str(tmp_path / "snps.vcf"), g, s)
    synth.write_fastq(str(tmp_path / "reads.fq"), r)
    for fn in ("ref.fa", "snps.vcf", "reads.fq"):
        assert _sha(str(tmp_path / fn)) == want[fn], fn


def test_oracle_reproduces_reference_vcf_on_ftiny(ftiny_dir, ftiny_reads):
    ix = O.OracleIndex.load(os.path.join(ftiny_dir, "idx"))
    r = ftiny_reads
    assert ix.process(r.bases, r.quals, r.offsets) == 0
    mine = O.calls_by_key(ix.sites(), index_io.read_chrlens(os.path.join(ftiny_dir, "idx.chrlens")))
    ref = O.parse_vcf_calls(os.path.join(GOLDEN, "ftiny.out.vcf.gz"))
    assert len(ref) == 2617
    assert mine == ref
    st = ix.stats.as_dict()
    # the fixture exercises every branch of the path
    for k in ("reads_n", "gate_open", "large_block", "refbf_pos", "snpbf_pos", "scan_ref", "scan_snp", "scan_oob",
              "aux_ref", "aux_snp", "site_test", "walks", "incr"):
        assert st[k] > 0, k
    assert st["reads"] == r.n and st["passes"] > st["reads"] - st["reads_n"]


def test_fixture_discriminates_bug_b1(ftiny_dir, ftiny_reads):
    """With the strided scan 'fixed' (stride 1) the calls change: the fixture sees B1."""
    ix = O.OracleIndex.load(os.path.join(ftiny_dir, "idx"))
    ix.set_scan_stride(1, 1)
    r = ftiny_reads
    ix.process(r.bases, r.quals, r.offsets)
    mine = O.calls_by_key(ix.sites(), index_io.read_chrlens(os.path.join(ftiny_dir, "idx.chrlens")))
    ref = O.parse_vcf_calls(os.path.join(GOLDEN, "ftiny.out.vcf.gz"))
    assert mine != ref


def test_oracle_threads_and_batches_are_order_independent(ftiny_dir, ftiny_reads):
    r = ftiny_reads
    a = O.OracleIndex.load(os.path.join(ftiny_dir, "idx"))
    a.process(r.bases, r.quals, r.offsets)
    sa = a.sites()
    b = O.OracleIndex.load(os.path.join(ftiny_dir, "idx"))
    half = r.n // 2
    for lo, hi in ((half, r.n), (0, half)):
        s = r.slice(lo, hi)
        b.process(s.bases, s.quals, s.offsets, nthreads=4)
    sb = b.sites()
    assert np.array_equal(sa["ref_cnt"], sb["ref_cnt"]) and np.array_equal(sa["alt_cnt"], sb["alt_cnt"])
    assert a.stats.as_dict() == b.stats.as_dict()


def test_caller_known_answers():
    """test/expected_output of the reference: saturated sites print GQ 846 (0/0 at 63/0, 1/1 at 0/63)."""
    assert O.call(63, 0, 253, 1)[0::2] == (1, 846)
    assert O.call(0, 63, 253, 1)[0::2] == (2, 846)
    assert O.call(0, 0, 127, 127)[0] == 0 and O.call(63, 63, 127, 127)[0] == 0
    g, conf, gq = O.call(5, 4, 127, 127)
    assert g == 3 and gq == int(-10 * np.log(conf))


def test_edge_reads(ftiny_dir):
    """Empty, shorter-than-32, N-containing and invalid reads (qv.cc:778-779, 815-828; util.c:103)."""
    ix = O.OracleIndex.load(os.path.join(ftiny_dir, "idx"))
    reads = [b"", b"ACGT", b"A" * 31, b"ACGTN" * 10, b"ACGT" * 8 + b"N", b"ACGX" * 8, b"acgt" * 16]
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8)
    offs = np.cumsum([0] + [len(x) for x in reads]).astype(np.uint64)
    quals = np.full(len(bases), ord("#"), np.uint8)
    assert ix.process(bases, quals, offs) == 1          # exactly the ACGX read is invalid
    st = ix.stats.as_dict()
    assert st["reads"] == 7 and st["reads_n"] == 1 and st["reads_invalid"] == 1
    # "ACGT"*8+"N": the N sits in the dropped tail (33rd base) -> the read is processed, not skipped
    assert st["chunks"] >= 1


def test_oracle_reproduces_reference_vcf_on_fsmall(tmp_path):
    """The second pin (24 711 genotyped records): F-small is the fixture with >= 100-entry reference buckets, dense HI24
    buckets, POS_AMBIGUOUS k-mers and 973 strided-scan reads past the end of the arrays.  Its index is written by the
    product's `vargeno index` and must first match the sha256 of the files the REFERENCE wrote for the same inputs
    (tests/golden/fsmall.sha256), so the oracle sees exactly the reference's index bytes."""
    import subprocess

    from conftest import BIN

    g, s, r = synth.f_small()
    d = str(tmp_path)
    synth.write_fasta(os.path.join(d, "ref.fa"), g)
    synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"),
                          stdout=subprocess.DEVNULL)
    want = read_sha256_list("fsmall")
    for fn in ("idx.chrlens", "idx.ref.dict", "idx.snp.dict", "idx.ref.bf", "idx.snp.bf"):
        assert _sha(os.path.join(d, fn)) == want[fn], fn
    ix = O.OracleIndex.load(os.path.join(d, "idx"))
    assert ix.process(r.bases, r.quals, r.offsets) == 0
    mine = O.calls_by_key(ix.sites(), index_io.read_chrlens(os.path.join(d, "idx.chrlens")))
    ref = O.parse_vcf_calls(os.path.join(GOLDEN, "fsmall.out.vcf.gz"))
    assert len(ref) == 24711
    assert mine == ref
    st = ix.stats.as_dict()
    assert st["large_block"] > 0 and st["scan_oob"] > 0 and st["aux_ref"] > 0 and st["aux_snp"] > 0
    # B1 discrimination on this fixture too: stride 1 changes thousands of calls
    ix.set_scan_stride(1, 1)
    ix.reset()
    ix.process(r.bases, r.quals, r.offsets)
    fixed = O.calls_by_key(ix.sites(), index_io.read_chrlens(os.path.join(d, "idx.chrlens")))
    assert sum(1 for k in ref if fixed.get(k) != ref[k]) > 1000


def test_oracle_reproduces_reference_vcf_on_fdense(tmp_path):
    """Third pin: the dense-bucket fixture (synth.f_dense: ~150 SNP k-mers per HI24 bucket, the shape of hg38 + full dbSNP,
    where iterate_snp_dict's strided scan walks hundreds of entries per gate-open chunk).  Index by the product's `vargeno
    index`, proven byte-identical to the reference's files first; then GT and GQ of every record the reference called."""
    import subprocess

    from conftest import BIN

    g, s, r = synth.f_dense()
    d = str(tmp_path)
    synth.write_fasta(os.path.join(d, "ref.fa"), g)
    synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
    want = read_sha256_list("fdense")
    for fn in ("ref.fa", "snps.vcf", "idx.chrlens", "idx.ref.dict", "idx.snp.dict", "idx.ref.bf", "idx.snp.bf"):
        assert _sha(os.path.join(d, fn)) == want[fn], fn
    ix = O.OracleIndex.load(os.path.join(d, "idx"))
    assert ix.process(r.bases, r.quals, r.offsets, nthreads=4) == 0
    mine = O.calls_by_key(ix.sites(), index_io.read_chrlens(os.path.join(d, "idx.chrlens")))
    ref = O.parse_vcf_calls(os.path.join(GOLDEN, "fdense.out.vcf.gz"))
    assert len(ref) > 10_000
    assert mine == ref
    st = ix.stats.as_dict()
    assert st["scan_snp"] > 100 * st["gate_open"]              # the scans are two orders of magnitude longer than on F-tiny
    assert st["large_block"] > 0 and st["aux_snp"] > 0


def test_oracle_caller_equals_the_reference_function_over_its_whole_domain():
    """tests/golden/caller_table.npz holds the REFERENCE's own choose_best_genotype (qv.cc:1789-1848; evaluated by
    oracle/ref_caller_table.cc, which #includes qv.cc) for every (ref_cnt, alt_cnt) in [0, 63]^2 and 16 allele-frequency
    pairs, with the GQ of qv.cc:1681: the oracle's caller must return the same genotype, the same GQ and the same confidence
    bits for all 65 536 of them."""
    z = np.load(os.path.join(GOLDEN, "caller_table.npz"))
    t = {k: z[k] for k in z.files}                     # (an NpzFile decompresses a member on every access)
    n = len(t["genotype"])
    assert n == 16 * 64 * 64
    bad = 0
    for i in range(n):
        g, conf, gq = O.call(t["ref_cnt"][i], t["alt_cnt"][i], t["ref_freq"][i], t["alt_freq"][i])
        ok = g == t["genotype"][i] and (g == 0 or (gq == t["gq"][i] and conf == t["conf"][i]))
        bad += not ok
    assert bad == 0


def test_oracle_vote_equals_the_reference_state_machine():
    """tests/golden/vote_table.npz: 2 400 seeded sequences of votes (ties, ambiguity flips, positions sharing a slot of the
    reference's table, refused neighbour votes, repeated k-mer positions, runs past 255 votes) and what the REFERENCE's own
    improved_index_table_add (qv.cc:132-178, driven by oracle/ref_vote_replay.cc) holds after each.  The oracle's vote must
    end every sequence with the same best position, the same uint8_t frequency and the same ambiguity flag."""
    z = np.load(os.path.join(GOLDEN, "vote_table.npz"))
    lens, index, kpos, neigh, want = (z[k] for k in ("lens", "index", "kpos", "neigh", "result"))
    ends = np.cumsum(lens.astype(np.int64))
    wrapped = 0
    for s in range(len(lens)):
        a, b = int(ends[s] - lens[s]), int(ends[s])
        got = O.vote_replay(index[a:b], kpos[a:b], neigh[a:b])
        w = tuple(int(x) for x in want[s])
        assert got[0] == w[0] and got[3] == w[3] and (not w[0] or got[1:3] == w[1:3]), (s, got, w)
        if w[0]:
            opened = np.nonzero((index[a:b] == w[1]) & (neigh[a:b] == 0))[0]
            votes = int((index[a + int(opened[0]):b] == w[1]).sum())
            wrapped += votes > 255 and w[2] == votes % 256
    assert int(want[:, 0].sum()) > 2000 and int(want[:, 3].sum()) > 100 and wrapped > 50


def test_reference_written_dictionaries_satisfy_what_the_loader_checks(ftiny_dir):
    """vg_index_open refuses dictionaries whose k-mers are not strictly increasing or whose "several positions" entries name
    auxiliary rows beyond the table.  The files the REFERENCE wrote (committed F-tiny index) satisfy both, so the check
    cannot turn away an index it should take."""
    rd = index_io.read_ref_dict(os.path.join(ftiny_dir, "idx.ref.dict"))
    sd = index_io.read_snp_dict(os.path.join(ftiny_dir, "idx.snp.dict"))
    for kmer, pos, amb, n_aux in ((rd["ref_kmer"], rd["ref_pos"], rd["ref_amb"], len(rd["ref_aux"])),
                                  (sd["snp_kmer"], sd["snp_pos"], sd["snp_amb"], len(sd["snp_aux_pos"]))):
        assert np.all(kmer[1:] > kmer[:-1])
        multi = (amb != 0) & (pos != 0xFFFFFFFF)
        assert multi.sum() > 0 and np.all(pos[multi] < n_aux)


def test_oracle_reproduces_reference_vcf_on_fstrands(tmp_path):
    """Fourth pin: the strand corner cases (synth.f_strands) the device's canonical-key views have to get right -- reference
    and SNP k-mers that are their own reverse complement (at one position and at two), k-mers whose reverse complement is in
    the dictionary as well, reads of both strands aimed at them, exact and one substitution away.  The fixture must hold those
    cases (counted from the dictionary files), the product's index must be the reference's byte for byte, and the oracle's
    calls the reference's."""
    import subprocess

    from conftest import BIN

    g, s, r, plants = synth.f_strands()
    d = str(tmp_path)
    synth.write_fasta(os.path.join(d, "ref.fa"), g)
    synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
    want = read_sha256_list("fstrands")
    for fn in ("ref.fa", "snps.vcf", "idx.chrlens", "idx.ref.dict", "idx.snp.dict", "idx.ref.bf", "idx.snp.bf"):
        assert _sha(os.path.join(d, fn)) == want[fn], fn
    synth.write_fastq(os.path.join(d, "reads.fq"), r)
    assert _sha(os.path.join(d, "reads.fq")) == want["reads.fq"]

    def revcomp(k):                                                  # 32-mers, two bits a base: complement, reverse the fields
        k = ~k.astype(np.uint64)
        for sh, mask in ((2, 0x3333333333333333), (4, 0x0F0F0F0F0F0F0F0F), (8, 0x00FF00FF00FF00FF), (16, 0x0000FFFF0000FFFF)):
            k = ((k >> np.uint64(sh)) & np.uint64(mask)) | ((k & np.uint64(mask)) << np.uint64(sh))
        return (k >> np.uint64(32)) | (k << np.uint64(32))

    rk = index_io.read_ref_dict(os.path.join(d, "idx.ref.dict"))["ref_kmer"]
    sk = index_io.read_snp_dict(os.path.join(d, "idx.snp.dict"))["snp_kmer"]
    assert int((revcomp(rk) == rk).sum()) >= 2 * (len(plants) // 3)                 # self-complementary reference k-mers
    assert int(np.isin(revcomp(rk), rk).sum()) >= len(plants)                       # ... + pairs present on both strands
    assert int((revcomp(sk) == sk).sum()) >= len(plants) // 3 - 5                   # self-complementary SNP k-mers
    ix = O.OracleIndex.load(os.path.join(d, "idx"))
    assert ix.process(r.bases, r.quals, r.offsets, nthreads=4) == 0
    mine = O.calls_by_key(ix.sites(), index_io.read_chrlens(os.path.join(d, "idx.chrlens")))
    ref = O.parse_vcf_calls(os.path.join(GOLDEN, "fstrands.out.vcf.gz"))
    assert len(ref) > 8_000
    assert mine == ref
    # the sites under the planted self-complementary SNP k-mers are among the called ones
    called = {int(k.split("$")[1]) for k in ref}
    xs = [int(p) for p in s.pos if 2_350 <= p and (int(p) - 2_351) % 2_700 < 32]
    assert len(xs) >= len(plants) // 3 and sum(1 for p in xs if p in called) >= len(plants) // 3 - 5


def test_oracle_reproduces_reference_vcf_on_flowcomplex(tmp_path):
    """Fifth pin: synth.f_lowcomplex(1) -- microsatellites whose runs are their own reverse complement, hairpins, tandem and
    dispersed copies (auxiliary rows, POS_AMBIGUOUS), one SNP per 25 bases.  Index files byte-identical to the reference's,
    then GT and GQ of every record it called.  (Seeds 1-15 were compared the same way when the fixture was made.)"""
    import subprocess

    from conftest import BIN

    g, s, r = synth.f_lowcomplex(1)
    d = str(tmp_path)
    synth.write_fasta(os.path.join(d, "ref.fa"), g)
    synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
    want = read_sha256_list("flowcomplex")
    for fn in ("ref.fa", "snps.vcf", "idx.chrlens", "idx.ref.dict", "idx.snp.dict", "idx.ref.bf", "idx.snp.bf"):
        assert _sha(os.path.join(d, fn)) == want[fn], fn
    ix = O.OracleIndex.load(os.path.join(d, "idx"))
    assert ix.process(r.bases, r.quals, r.offsets, nthreads=4) == 0
    mine = O.calls_by_key(ix.sites(), index_io.read_chrlens(os.path.join(d, "idx.chrlens")))
    ref = O.parse_vcf_calls(os.path.join(GOLDEN, "flowcomplex.out.vcf.gz"))
    assert len(ref) == 2082
    assert mine == ref
    st = ix.stats.as_dict()
    rd = index_io.read_ref_dict(os.path.join(d, "idx.ref.dict"))
    assert st["aux_ref"] > 2000 and st["scan_oob"] > 0 and int((rd["ref_pos"] == 0xFFFFFFFF).sum()) > 500
    for fn in ("idx.ref.bf", "idx.snp.bf"):
        os.remove(os.path.join(d, fn))


def test_oracle_reproduces_reference_vcf_on_frepeated(tmp_path):
    """synth.f_repeated_records: an SNP list that holds the same record up to five times -- auxiliary rows listing ONE position
    several times, i.e. a chunk that votes, and walks the pile-up, several times for the same context (qv.cc:913-933, 1444-1494).
    The product's index files must be the reference's (sha256), and the oracle's calls the reference's VCF."""
    import subprocess

    from conftest import BIN

    q = synth.f_repeated_records()
    d = str(tmp_path)
    synth.write_quirk(d, q)
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
    want = read_sha256_list("frepeated")
    for fn in ("ref.fa", "snps.vcf", "reads.fq", "idx.chrlens", "idx.ref.dict", "idx.snp.dict", "idx.ref.bf", "idx.snp.bf"):
        assert _sha(os.path.join(d, fn)) == want[fn], fn
    r = q["reads"]
    ix = O.OracleIndex.load(os.path.join(d, "idx"))
    assert ix.process(r.bases, r.quals, r.offsets) == 0
    mine = O.calls_by_key(ix.sites(), index_io.read_chrlens(os.path.join(d, "idx.chrlens")))
    ref = O.parse_vcf_calls(os.path.join(GOLDEN, "frepeated.out.vcf.gz"))
    assert len(ref) >= 40
    assert mine == ref
    assert ix.stats.as_dict()["aux_snp"] > 500
    for fn in ("idx.ref.bf", "idx.snp.bf"):
        os.remove(os.path.join(d, fn))


def test_reference_binary_on_disk_is_what_the_committed_recipe_builds():
    """The reference binary that travels to the GPU box (oracle/_ref/vargeno: the cpu_baseline of record and the comparand of the
    command-line tests there) is the one `make -C oracle ref` produces from the recipe as committed: its sha256 is recorded in
    tests/golden/ref_binary.sha256 when the recipe or the reference changes, the recipe lists its own Makefile as a prerequisite
    (round 5's binary predated a change of -march and was never rebuilt), and bench.py prints the hash in cpu_baseline.sample."""
    import re
    import subprocess

    from conftest import ROOT

    mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    for target in (r"\$\(OUT\)/vargeno:", r"\$\(OUT\)/caller_table\.bin:", r"\$\(OUT\)/ref_vote_replay:"):
        rule = re.search(target + r"[^\n]*", mk).group(0)
        assert "$(THIS)" in rule, rule
    binp = os.path.join(ROOT, "oracle", "_ref", "vargeno")
    want = open(os.path.join(GOLDEN, "ref_binary.sha256")).read().split()[0]
    if os.path.isdir("/root/reference/src"):                        # the build container: make is a no-op when the binary is current
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(binp):
        import pytest

        pytest.skip("no oracle/_ref/vargeno here")
    assert _sha(binp) == want
    assert open(os.path.join(ROOT, "oracle", "_ref", "vargeno.sha256")).read().split()[0] == want
