"""Seeded synthetic genome / SNP list / donor / read generator.

Host-side data plumbing for tests and bench.py; nothing here is on the hot path.  Everything is a
function of one integer seed (numpy PCG64), so the GPU box regenerates the same bytes as this
container and the committed sha256 list in tests/golden/ pins them.

Conventions follow the reference's file formats (README.md:61-72 of the reference): FASTA with
70-column lines, VCF with 8 columns and ``CAF=ref,alt`` in INFO, 4-line FASTQ, Phred+33.
"""
from __future__ import annotations

import io
import os
from dataclasses import dataclass, field

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[_a] = _b
_CODE = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i
    _CODE[_c + 32] = _i


@dataclass
class Genome:
    names: list            # FASTA header text after '>' (may contain a description)
    seqs: list             # list of np.uint8 arrays (ASCII, upper-case ACGTN)

    @property
    def short_names(self):
        # fasta_parser.c:67-75 of the reference cuts the name at '|' / whitespace / 64 chars
        out = []
        for n in self.names:
            cut = len(n)
            for i, ch in enumerate(n):
                if ch == "|" or ch.isspace() or i == 64:
                    cut = i
                    break
            out.append(n[:cut])
        return out

    @property
    def total_len(self):
        return int(sum(len(s) for s in self.seqs))


@dataclass
class SnpSet:
    chrom: np.ndarray      # int32 index into genome.seqs
    pos: np.ndarray        # int64, 1-based within chromosome
    ref: np.ndarray        # uint8 ASCII
    alt: np.ndarray        # uint8 ASCII
    caf_ref: np.ndarray    # float64
    caf_alt: np.ndarray
    genotype: np.ndarray = field(default=None)   # donor: 0 = 0/0, 1 = 0/1, 2 = 1/1


@dataclass
class Reads:
    """Flat ASCII read batch, the layout the C-ABI takes: bases/quals concatenated, offsets[n+1]."""
    bases: np.ndarray      # uint8
    quals: np.ndarray      # uint8 (same offsets)
    offsets: np.ndarray    # uint64, n+1

    @property
    def n(self):
        return len(self.offsets) - 1

    def slice(self, lo, hi):
        o = self.offsets
        b0, b1 = int(o[lo]), int(o[hi])
        return Reads(self.bases[b0:b1], self.quals[b0:b1], (o[lo:hi + 1] - o[lo]).astype(np.uint64))


def random_bases(rng, n):
    return ACGT[rng.integers(0, 4, size=n, dtype=np.uint8)]


def make_genome(rng, lengths, names, *, repeats_per_mbp=50.0, repeat_len=(200, 2000), repeat_div=0.03,
                microsat_per_mbp=12.5, n_gaps=True, plant_block16=0, plant_hi24=0, plant_copy14=False,
                plant_ac_run=False, plant_t16=0, exact_repeat_frac=0.0):
    """i.i.d. uniform ACGT with planted structure (SURVEY.md §8c F-small / F-mid recipes).

    plant_block16: number of copies of one 16-mer, each preceded by 16 random bases, i.e. that many
                   distinct 32-mers sharing HI32 -> a ref-dict bucket >= BLOCK_SIZE_THRESHOLD (vartype.h:103).
    plant_hi24:    number of copies of one 12-mer (dense HI24 buckets in the SNP dict).
    plant_copy14:  one 100-mer copied 14x (> AUX_TABLE_COLS occurrences -> POS_AMBIGUOUS, dictgen.c:118).
    plant_ac_run:  an (AC)200 microsatellite.
    plant_t16:     number of [16 random bases][T x 16] segments: 32-mers in the LAST ref-dict bucket
                   (HI32 = 0xFFFFFFFF) and SNP k-mers in the last SNP bucket, so the strided scan (B1)
                   runs past the end of the arrays (SURVEY.md §8a "out-of-range t").
    exact_repeat_frac: fraction of every sequence covered by planted repeat FAMILIES whose copies are (nearly) identical --
                   what drives the reference's auxiliary rows (2-10 occurrences of a 32-mer, dictgen.c:63-154) and
                   POS_AMBIGUOUS (> 10): 70 % of those bases in families of 2-10 copies of 300-3000 bp (segmental-
                   duplication-like: a third of the copies identical, the others 0.5 % or 2 % diverged), 30 % in families of
                   11-200 copies of 250-350 bp (young-interspersed-repeat-like, 0-5 % diverged).
    """
    seqs = []
    for L in lengths:
        s = random_bases(rng, L)
        mbp = L / 1e6
        # diverged repeats: copy a segment elsewhere with `repeat_div` substitutions
        for _ in range(int(round(repeats_per_mbp * mbp))):
            rl = int(rng.integers(repeat_len[0], repeat_len[1]))
            if rl * 2 + 200 >= L:
                continue
            src = int(rng.integers(0, L - rl))
            dst = int(rng.integers(0, L - rl))
            seg = s[src:src + rl].copy()
            m = rng.random(rl) < repeat_div
            seg[m] = random_bases(rng, int(m.sum()))
            s[dst:dst + rl] = seg
        # repeat families of (nearly) identical copies
        if exact_repeat_frac > 0:
            for share, clo, chi, llo, lhi, divs in ((0.7, 2, 10, 300, 3000, (0.0, 0.005, 0.02)), (0.3, 11, 200, 250, 350, (0.0, 0.01, 0.05))):
                budget = int(exact_repeat_frac * share * L)
                while budget > 0:
                    rl = int(rng.integers(llo, lhi + 1))
                    copies = int(rng.integers(clo, chi + 1))
                    if rl * 2 + 200 >= L:
                        break
                    unit = s[int(rng.integers(0, L - rl)):][:rl].copy()
                    for dst in rng.integers(0, L - rl, size=copies):
                        seg = unit
                        dv = divs[int(rng.integers(0, len(divs)))]
                        if dv > 0:
                            seg = unit.copy()
                            m = rng.random(rl) < dv
                            seg[m] = random_bases(rng, int(m.sum()))
                        s[int(dst):int(dst) + rl] = seg
                    budget -= rl * copies
        # microsatellites
        for _ in range(int(round(microsat_per_mbp * mbp))):
            unit = random_bases(rng, int(rng.integers(1, 5)))
            reps = int(rng.integers(15, 60))
            run = np.tile(unit, reps)
            if len(run) + 100 >= L:
                continue
            dst = int(rng.integers(0, L - len(run)))
            s[dst:dst + len(run)] = run
        seqs.append(s)
    s0 = seqs[0]
    L0 = len(s0)
    if plant_block16:
        mer = random_bases(rng, 16)
        for d in rng.choice(np.arange(64, L0 - 64, 48), size=plant_block16, replace=False):
            s0[d + 16:d + 32] = mer
    if plant_hi24:
        mer = random_bases(rng, 12)
        for d in rng.choice(np.arange(64, L0 - 64, 40), size=plant_hi24, replace=False):
            s0[d:d + 12] = mer
    if plant_copy14:
        mer = random_bases(rng, 100)
        for d in rng.choice(np.arange(200, L0 - 200, 128), size=14, replace=False):
            s0[d:d + 100] = mer
    if plant_ac_run:
        d = int(rng.integers(1000, L0 - 1000))
        s0[d:d + 400] = np.tile(np.frombuffer(b"AC", dtype=np.uint8), 200)
    if plant_t16:
        for d in rng.choice(np.arange(600, L0 - 600, 64), size=plant_t16, replace=False):
            s0[d + 16:d + 32] = ord("T")
    if n_gaps:
        for s in seqs:
            g = min(500, len(s) // 100)
            s[:g] = ord("N")                       # gap at chromosome start
        g = min(300, L0 // 100)
        mid = L0 // 2
        s0[mid:mid + g] = ord("N")                 # one interior gap
    return Genome(list(names), seqs)


def make_snps(rng, genome, n, *, with_caf=True, genotypes="uniform"):
    """Uniform positions (not N, not within 32 of a chromosome end), one alt != ref, CAF ~ Beta(5,1).
    Donor genotypes: "uniform" over {0/0, 0/1, 1/1}, or "hwe": each haplotype carries the alt allele with
    the SNP's alt frequency -- the dense lists (100 M SNPs, one per 31 bp) need that to keep k-mers with a single alt common."""
    lens = np.array([len(s) for s in genome.seqs], dtype=np.int64)
    total = int(lens.sum())
    n = min(n, total // 3)
    if total <= 1 << 28:
        glob = np.sort(rng.choice(total, size=n, replace=False))
    else:                                        # a permutation of a 3 Gbp population is too slow: draw, de-duplicate, trim
        glob = np.unique(rng.integers(0, total, size=int(n * 1.02) + 1000, dtype=np.int64))
        if len(glob) > n:
            glob = np.sort(rng.choice(glob, size=n, replace=False))
    starts = np.concatenate([[0], np.cumsum(lens)])
    chrom = (np.searchsorted(starts, glob, side="right") - 1).astype(np.int32)
    pos0 = glob - starts[chrom]
    cat = np.concatenate(genome.seqs)
    ref = cat[glob]
    keep = (ref != ord("N")) & (pos0 >= 40) & (pos0 + 40 < lens[chrom])
    chrom, pos0, ref = chrom[keep], pos0[keep], ref[keep]
    refc = _CODE[ref]
    altc = (refc + rng.integers(1, 4, size=len(refc), dtype=np.uint8)) % 4
    p = 0.01 + 0.98 * rng.beta(5.0, 1.0, size=len(refc))
    s = SnpSet(chrom, pos0 + 1, ref, ACGT[altc], p, 1.0 - p)
    if genotypes == "hwe":
        q = 1.0 - p
        s.genotype = (rng.random(len(refc)) < q).astype(np.uint8) + (rng.random(len(refc)) < q).astype(np.uint8)    # 0 = 0/0, 1 = 0/1, 2 = 1/1
    else:
        s.genotype = rng.integers(0, 3, size=len(refc)).astype(np.uint8)
    s._with_caf = with_caf
    return s


def haplotypes(genome, snps):
    """Two donor haplotypes of the concatenated genome (hap0 carries alt only for 1/1)."""
    cat = np.concatenate(genome.seqs)
    lens = np.array([len(s) for s in genome.seqs], dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(lens)])
    g = starts[snps.chrom] + snps.pos - 1
    h0, h1 = cat.copy(), cat.copy()
    m1 = snps.genotype >= 1
    m2 = snps.genotype == 2
    h1[g[m1]] = snps.alt[m1]
    h0[g[m2]] = snps.alt[m2]
    return h0, h1, starts


def make_reads(rng, genome, snps, n, *, lengths=(150,), err=0.005, lowq=0.08, lower_frac=0.0,
               rev_frac=0.5):
    """n reads; uniform start, uniform haplotype, `rev_frac` reverse strand, substitution errors,
    Phred+33 qualities: with prob `lowq` uniform in '#'..'7' (gate-open for '8' = QUALITY_SCORE,
    vartype.h:17) else uniform in ':'..'I'."""
    h0, h1, starts = haplotypes(genome, snps)
    G = len(h0)
    lens_choice = np.asarray(lengths, dtype=np.int64)
    rl = lens_choice[rng.integers(0, len(lens_choice), size=n)]
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(rl)
    total = int(offsets[-1])
    bases = np.empty(total, dtype=np.uint8)
    for L in np.unique(rl):
        idx = np.nonzero(rl == L)[0]
        m = len(idx)
        L = int(L)
        st = rng.integers(0, G - L, size=m)
        gi = st[:, None] + np.arange(L)[None, :]
        hap = rng.integers(0, 2, size=m).astype(bool)
        b = np.where(hap[:, None], h1[gi], h0[gi])
        e = rng.random((m, L)) < err
        ne = int(e.sum())
        if ne:
            cur = _CODE[b[e]]
            sub = (cur + rng.integers(1, 4, size=ne, dtype=np.uint8)) % 4   # N (code 4) -> some base; harmless
            b[e] = ACGT[sub]
        rv = rng.random(m) < rev_frac
        b[rv] = _COMP[b[rv][:, ::-1]]
        dst = offsets[idx].astype(np.int64)[:, None] + np.arange(L)[None, :]
        bases[dst] = b
    q_hi = rng.integers(ord(":"), ord("I") + 1, size=total, dtype=np.uint8)
    q_lo = rng.integers(ord("#"), ord("7") + 1, size=total, dtype=np.uint8)
    quals = np.where(rng.random(total) < lowq, q_lo, q_hi).astype(np.uint8)
    if lower_frac > 0:
        lc = rng.random(total) < lower_frac
        bases[lc] |= 0x20
    return Reads(bases, quals, offsets)


# ----------------------------------------------------------------------------- writers

def write_fasta(path, genome, width=70, softmask=0.0, seed=20261002):
    """softmask: fraction of every sequence written in LOWER case, in runs of 300-3 000 bases (what RepeatMasker leaves in a UCSC
    download).  The dictionary pass of `vargeno index` folds case (fasta_parser.c), its bit-vector pass does not
    (generate_bf.cc:230): an index from such a file has a reference bit vector that is NOT the set of the dictionary's LO32 values."""
    rng = np.random.default_rng(seed + 77) if softmask > 0 else None
    with open(path, "wb") as f:
        for name, s in zip(genome.names, genome.seqs):
            f.write(b">" + name.encode() + b"\n")
            n = len(s)
            if rng is not None and n:
                k = max(1, int(softmask * n / 1650))
                starts = rng.integers(0, n, k)
                lens = rng.integers(300, 3000, k)
                d = np.zeros(n + 1, np.int32)
                np.add.at(d, starts, 1)
                np.add.at(d, np.minimum(starts + lens, n), -1)
                s = np.where(np.cumsum(d[:n]) > 0, s | 0x20, s).astype(np.uint8)
            full = (n // width) * width
            if full:
                body = np.empty((full // width, width + 1), dtype=np.uint8)
                body[:, :width] = s[:full].reshape(-1, width)
                body[:, width] = 10
                f.write(body.tobytes())
            if n > full:
                f.write(s[full:].tobytes() + b"\n")


VCF_HEADER = (
    "##fileformat=VCFv4.0\n"
    "##source=vargeno_amd.synth\n"
    "##INFO=<ID=CAF,Number=.,Type=String,Description=\"ref,alt allele frequencies\">\n"
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
)


def _decimal_columns(v, width):
    """uint8[len(v), width]: decimal digits of v right-aligned, 0 bytes in front (dropped when the lines are assembled)."""
    v = np.asarray(v, dtype=np.int64)
    out = np.zeros((len(v), width), dtype=np.uint8)
    rest = v.copy()
    for k in range(width - 1, -1, -1):
        live = (rest > 0) | (k == width - 1)          # the last digit is always written (the value 0 prints as "0")
        out[live, k] = (48 + rest % 10)[live]
        rest //= 10
    return out


def _write_vcf_vectorised(path, vnames, snps, with_caf):
    """Same lines as the loop in write_vcf() -- canonical decimals, "%.4f" frequencies -- assembled as byte columns."""
    n = len(snps.pos)
    name_w = max(len(v) for v in vnames)
    names = np.zeros((len(vnames), name_w), dtype=np.uint8)
    for i, v in enumerate(vnames):
        names[i, :len(v)] = np.frombuffer(v.encode(), dtype=np.uint8)
    ids = np.arange(1, n + 1, dtype=np.int64)

    def lit(text):
        return np.broadcast_to(np.frombuffer(text.encode(), dtype=np.uint8), (n, len(text)))

    def freq(x):                                     # "%.4f" of a value in [0, 10): d.dddd
        q = np.floor(np.asarray(x, dtype=np.float64) * 10000.0 + 0.5).astype(np.int64)
        c = np.empty((n, 6), dtype=np.uint8)
        c[:, 0] = 48 + q // 10000
        c[:, 1] = ord(".")
        for k in range(4):
            c[:, 2 + k] = 48 + (q // 10 ** (3 - k)) % 10
        return c

    cols = [names[snps.chrom], lit("\t"), _decimal_columns(snps.pos, 10), lit("\trs"), _decimal_columns(ids, 9), lit("\t"),
            snps.ref.reshape(-1, 1), lit("\t"), snps.alt.reshape(-1, 1), lit("\t.\t.\tRS="), _decimal_columns(ids, 9)]
    if with_caf:
        cols += [lit(";CAF="), freq(snps.caf_ref), lit(","), freq(snps.caf_alt), lit(";COMMON=1\n")]
    else:
        cols += [lit("\n")]
    with open(path, "wb") as f:
        f.write(VCF_HEADER.encode())
        step = 1 << 20
        for lo in range(0, n, step):
            blk = np.concatenate([c[lo:lo + step] for c in cols], axis=1).reshape(-1)
            f.write(blk[blk != 0].tobytes())


def write_vcf(path, genome, snps, *, strip_chr=True):
    names = genome.short_names
    vnames = [n[3:] if (strip_chr and n.startswith("chr")) else n for n in names]
    with_caf = getattr(snps, "_with_caf", True)
    if len(snps.pos) > 200_000:                      # the bench-scale lists: millions of lines
        return _write_vcf_vectorised(path, vnames, snps, with_caf)
    buf = io.StringIO()
    buf.write(VCF_HEADER)
    for i in range(len(snps.pos)):
        info = ("RS=%d;CAF=%.4f,%.4f;COMMON=1" % (i + 1, snps.caf_ref[i], snps.caf_alt[i])) if with_caf else "RS=%d" % (i + 1)
        buf.write("%s\t%d\trs%d\t%c\t%c\t.\t.\t%s\n" % (vnames[snps.chrom[i]], snps.pos[i], i + 1,
                                                      snps.ref[i], snps.alt[i], info))
    with open(path, "w") as f:
        f.write(buf.getvalue())


def write_fastq(path, reads):
    o = reads.offsets.astype(np.int64)
    with open(path, "wb") as f:
        chunk = []
        for i in range(reads.n):
            chunk.append(b"@r%d\n" % i)
            chunk.append(reads.bases[o[i]:o[i + 1]].tobytes())
            chunk.append(b"\n+\n")
            chunk.append(reads.quals[o[i]:o[i + 1]].tobytes())
            chunk.append(b"\n")
            if len(chunk) >= 50000:
                f.write(b"".join(chunk))
                chunk = []
        f.write(b"".join(chunk))


# ----------------------------------------------------------------------------- named data sets

def f_small(seed=20261002):
    """F-small: 300 kbp, 2 chromosomes, adversarial plants, 30 k SNPs, 40 k reads."""
    rng = np.random.default_rng(seed)
    g = make_genome(rng, [220_000, 80_000], ["chr22 synthetic", "chrX"], repeats_per_mbp=200.0,
                    repeat_len=(200, 1500), repeat_div=0.03, microsat_per_mbp=40.0,
                    plant_block16=160, plant_hi24=400, plant_copy14=True, plant_ac_run=True)
    s = make_snps(rng, g, 30_000)
    r = make_reads(rng, g, s, 40_000, lengths=(150, 150, 150, 101, 250, 64, 31), err=0.01, lowq=0.40,
                   lower_frac=0.05)
    return g, s, r


def f_dense(seed=424242):
    """Dense-bucket fixture: the bucket shapes of BASELINE.json configs[4] (hg38 + full dbSNP: ~190 SNP-dictionary entries
    per HI24 bucket) on a genome small enough for the reference to index in seconds.  Each chromosome is random over a
    TWO-letter alphabet ({A,C} / {G,T}), so only 2^12 distinct 12-mers end its k-mers: 60 k SNPs (one base in ten) put
    ~150 k-mers into each of those SNP buckets, reference buckets (HI32: 2^16 distinct 16-mers) hold ~5 entries, the
    LO32-ordered view is as dense, and the reverse complement of a chr1 read looks like chr2 text.  The strided bucket
    scans (B1, qv.cc:316-376, 413-464) walk hundreds of entries per gate-open chunk here."""
    rng = np.random.default_rng(seed)
    lens = [360_000, 240_000]
    seqs = []
    for L, ab in zip(lens, (b"AC", b"GT")):
        letters = np.frombuffer(ab, dtype=np.uint8)
        seqs.append(letters[rng.integers(0, 2, size=L, dtype=np.uint8)])
    s0 = seqs[0]
    mer = random_bases(rng, 16)
    for d in rng.choice(np.arange(64, len(s0) - 64, 48), size=130, replace=False):      # one reference bucket of >= 100 entries
        s0[d + 16:d + 32] = mer
    for sq in seqs:
        sq[:200] = ord("N")
    g = Genome(["chr1", "chr2"], seqs)
    s = make_snps(rng, g, 60_000)
    r = make_reads(rng, g, s, 15_000, lengths=(150, 150, 101, 250), err=0.01, lowq=0.30)
    return g, s, r


def f_tiny(seed=7):
    """A few-second data set for CPU unit tests: 60 kbp, 3 k SNPs, 4 k reads."""
    rng = np.random.default_rng(seed)
    g = make_genome(rng, [40_000, 20_000], ["chr1", "chr2 second"], repeats_per_mbp=300.0,
                    repeat_len=(100, 600), microsat_per_mbp=100.0, plant_block16=120, plant_hi24=150,
                    plant_copy14=True, plant_ac_run=True, plant_t16=60)
    s = make_snps(rng, g, 3_000)
    r = make_reads(rng, g, s, 4_000, lengths=(150, 150, 101, 64, 31, 250), err=0.01, lowq=0.40,
                   lower_frac=0.05)
    return g, s, r



def f_manykeys(seed=5, n_homes=6, copies_per_home=8, n_random_reads=4_000):
    """Reads that outgrow the DEEP wave tier (48 vote keys) and are finished by the lane machine: a 400 kbp genome in which seven
    unrelated 32-mers A0..A6 stand side by side at a "home" and each of them nine more times at scattered places -- ten copies,
    the most a dictionary entry lists before it turns POS_AMBIGUOUS (dictgen.c:118) --, so that a 224-base read of the home finds
    ten positions per chunk: one vote key for the home, 7 x 9 = 63 for the scattered copies.  `n_homes` such homes, every home
    read `copies_per_home` times (both strands, some with an error in a low-quality chunk), among ordinary reads.
    Returns (genome, snps, reads)."""
    rng = np.random.default_rng(seed)
    g = make_genome(rng, [400_000], ["chr1"], repeats_per_mbp=10.0, microsat_per_mbp=0.0, n_gaps=False)
    s0 = g.seqs[0]
    homes = 10_000 + 60_000 * np.arange(n_homes, dtype=np.int64)
    for hi, H in enumerate(homes):
        H = int(H)
        for c in range(7):
            a = s0[H + 32 * c:H + 32 * c + 32].copy()
            for j in range(9):
                at = H + 1_000 + 523 * (9 * c + j) + hi                  # scattered, never side by side with another planted k-mer
                s0[at:at + 32] = a
    s = make_snps(rng, g, 4_000)
    r = make_reads(rng, g, s, n_random_reads, lengths=(150, 250, 224), err=0.01, lowq=0.3)
    h0, _, _ = haplotypes(g, s)
    extra_b, extra_q = [], []
    for H in homes:
        for k in range(copies_per_home):
            b = h0[int(H):int(H) + 224].copy()
            q = rng.integers(ord(":"), ord("I") + 1, size=224, dtype=np.uint8)
            if k % 4 == 3:                                               # an error in chunk 2, whose quality character says "look for neighbours"
                b[70] = ACGT[(_CODE[b[70]] + 1) % 4]
                q[2] = ord("#")
            if k % 2:
                b = _COMP[b[::-1]]
            extra_b.append(b)
            extra_q.append(q)
    nb = np.concatenate([r.bases] + extra_b)
    nq = np.concatenate([r.quals] + extra_q)
    no = np.concatenate([r.offsets, r.offsets[-1] + np.uint64(224) * np.arange(1, len(extra_b) + 1, dtype=np.uint64)])
    return g, s, Reads(nb, nq, no.astype(np.uint64))


def f_strands(seed=31, n_plants=300, n_random_reads=12_000):
    """Strand corner cases of a canonical-key index (one look-up of min(K, revcomp K) answers both strands): a 300 kbp genome
    with planted 32-mers that ARE their own reverse complement (X + revcomp X; once, and at two positions), and 32-mers whose
    reverse complement also occurs on the forward strand, SNPs through and around them, and reads -- both strands, both
    haplotypes -- placed so that one of their 32-base chunks is exactly a planted 32-mer, a third of them with a substitution
    inside that chunk (the Hamming-1 search then has the planted k-mer as a neighbour), 40 % gate-open chunks; and SNP k-mers
    that are their own reverse complement (the alt allele completes the palindrome).
    Returns (genome, snps, reads, plant_positions)."""
    rng = np.random.default_rng(seed)
    g = make_genome(rng, [300_000], ["chr1"], repeats_per_mbp=20.0, microsat_per_mbp=0.0, n_gaps=False)
    s0 = g.seqs[0]
    plants = 2_000 + 900 * np.arange(n_plants, dtype=np.int64)
    for i, p in enumerate(plants):
        p = int(p)
        if i % 3 == 1:                                               # K here, revcomp K 350 bases on: both on the forward strand
            s0[p + 350:p + 382] = _COMP[s0[p:p + 32][::-1]]
            continue
        x = random_bases(rng, 16)
        s0[p:p + 16] = x
        s0[p + 16:p + 32] = _COMP[x[::-1]]                           # K == revcomp K
        if i % 3 == 2:
            s0[p + 350:p + 382] = s0[p:p + 32]                       # ... at two positions
    s = make_snps(rng, g, 9_000)
    # a SNP k-mer that is its own reverse complement: next to every single-position plant, the same 32-mer with one base changed
    # in the reference and a SNP there whose alt allele restores it (donor genotypes 0/1 and 1/1 alternate)
    xp, xref, xalt = [], [], []
    for i, p in enumerate(plants[0::3]):
        p = int(p)
        j = int(rng.integers(0, 32))
        s0[p + 350:p + 382] = s0[p:p + 32]
        xalt.append(s0[p + j])
        s0[p + 350 + j] = ACGT[(_CODE[s0[p + j]] + 1 + i % 3) % 4]
        xref.append(s0[p + 350 + j])
        xp.append(p + 350 + j + 1)
    keep = ~np.isin(s.pos, xp)
    pos = np.concatenate([s.pos[keep], np.array(xp, dtype=np.int64)])
    order = np.argsort(pos, kind="stable")
    nx = len(xp)
    cat = lambda a, b, dt: np.concatenate([a[keep], np.asarray(b, dtype=dt)])[order]
    with_caf = s._with_caf
    s = SnpSet(cat(s.chrom, np.zeros(nx), np.int32), pos[order], cat(s.ref, xref, np.uint8), cat(s.alt, xalt, np.uint8),
               cat(s.caf_ref, np.full(nx, 0.6), np.float64), cat(s.caf_alt, np.full(nx, 0.4), np.float64),
               cat(s.genotype, 1 + np.arange(nx) % 2, np.uint8))
    s._with_caf = with_caf
    s.ref = np.concatenate(g.seqs)[s.pos - 1]                        # (a plant may have overwritten the base under a drawn SNP)
    clash = s.ref == s.alt
    s.alt[clash] = ACGT[(_CODE[s.ref[clash]] + 1) % 4]
    h0, h1, _ = haplotypes(g, s)
    L = 150
    starts, rev = [], []
    for p in plants:
        for q in (int(p), int(p) + 350):
            for c in range(4):
                starts += [q - 32 * c, q + 32 * (c + 1) - L]         # chunk c of the forward read / of the reverse read is [q, q+32)
                rev += [False, True]
    starts, rev = np.array(starts, dtype=np.int64), np.array(rev)
    starts, rev = np.tile(starts, 2), np.tile(rev, 2)
    hap = np.repeat([False, True], len(starts) // 2)
    m = len(starts)
    gi = starts[:, None] + np.arange(L)[None, :]
    b = np.where(hap[:, None], h1[gi], h0[gi])
    mut = np.nonzero(rng.random(m) < 1.0 / 3.0)[0]                   # one substitution somewhere in the read's planted chunk
    tgt = np.tile(np.repeat(np.repeat(plants, 2) + np.tile([0, 350], len(plants)), 8), 2)   # the planted interval each read was made for
    where = tgt[mut] - starts[mut] + rng.integers(0, 32, size=len(mut))
    cur = _CODE[b[mut, where]]
    b[mut, where] = ACGT[(cur + rng.integers(1, 4, size=len(mut), dtype=np.uint8)) % 4]
    b[rev] = _COMP[b[rev][:, ::-1]]
    total = m * L
    q_hi = rng.integers(ord(":"), ord("I") + 1, size=total, dtype=np.uint8)
    q_lo = rng.integers(ord("#"), ord("7") + 1, size=total, dtype=np.uint8)
    quals = np.where(rng.random(total) < 0.40, q_lo, q_hi).astype(np.uint8)
    placed = Reads(b.reshape(-1).copy(), quals, (np.arange(m + 1, dtype=np.uint64) * np.uint64(L)))
    rnd = make_reads(rng, g, s, n_random_reads, lengths=(150, 101, 250), err=0.01, lowq=0.40)
    reads = Reads(np.concatenate([placed.bases, rnd.bases]), np.concatenate([placed.quals, rnd.quals]),
                  np.concatenate([placed.offsets, rnd.offsets[1:] + placed.offsets[-1]]).astype(np.uint64))
    return g, s, reads, plants


def _fasta_norm(stream):
    """What the reference's FASTA reader makes of a sequence's characters (fasta_parser.c:7-25): ACGT in either case -> upper
    case, ANY other character that is not a newline -> 'N' (IUPAC codes, blanks, carriage returns all count as a base)."""
    out = np.full(len(stream), ord("N"), np.uint8)
    code = _CODE[stream]
    ok = code < 4
    out[ok] = ACGT[code[ok]]
    return out


def f_quirk(seed=99):
    """Inputs with the irregularities real FASTA / dbSNP files have, for the producer of the index (`vargeno index`,
    src/dictgen.c:561-794, src/fasta_parser.c) and the VCF pass of `geno`: a pure function of the seed.  Returns a dict:
    fasta (bytes), vcf (str), reads (Reads), genome (Genome as the reference reads it: short names, normalised bases).

    FASTA: names with '|' and descriptions, a name longer than 64 characters, lower-case (soft-masked) runs -- what the
    bit-vector pass compares case-sensitively (generate_bf.cc:230) and the dictionary pass does not --, N and n runs, lines
    of uneven width, empty lines, no newline at the end of the file.
    VCF: records out of order, multi-allelic and indel records, lower-case alleles, ALT equal to REF, REF 'N',
    chromosomes the FASTA does not have (one of them the 64-character name, which the 50-byte name buffer of dictgen.c:575
    cannot hold), names with and without "chr", positions at and beyond the 32-base margins, SNPs next to N runs,
    repeated positions, empty and comment lines in the body, CAF anywhere in INFO, a later INFO key that merely starts with
    "CAF" (dictgen.c:718-724 takes the last such key), '.' and exponent notation in CAF, extra columns.
    Left out because the reference aborts on them or its behaviour is undefined: characters other than ACGTN in the FASTA
    (IUPAC codes, blanks, carriage returns: generate_bf.cc:132 -> util.c:122 assert), a one-character ALT that is not a base
    ('.', 'N': generate_bf.cc:257, same assert), a sequence shorter than 32 bases, a processed record without CAF after one
    with it, a record beyond its chromosome's end or whose REF disagrees with the FASTA (fatal, dictgen.c:666-672)."""
    rng = np.random.default_rng(seed)
    long_name = "chr" + "L" * 80
    names = ["chr1|gi|555|ref|NC_1.1| Homo quirkus chromosome 1", "chrUn_gl000220 unplaced scaffold", "contig7", long_name + " too long"]
    g0 = make_genome(rng, [30_000, 14_000, 9_000, 5_000], names, repeats_per_mbp=300.0, repeat_len=(100, 500),
                     microsat_per_mbp=80.0, plant_block16=40, plant_hi24=40, plant_t16=20)
    streams = []
    for s in g0.seqs:
        s = s.copy()
        for _ in range(8):                                   # soft-masked runs
            a = int(rng.integers(0, len(s) - 600)); s[a:a + int(rng.integers(30, 500))] |= 0x20
        for _ in range(5):                                   # short N runs inside the sequence, some of them lower case
            a = int(rng.integers(1000, len(s) - 1000)); s[a:a + int(rng.integers(1, 40))] = ord("N") if rng.random() < 0.6 else ord("n")
        streams.append(s)
    norm = [_fasta_norm(s) for s in streams]
    g = Genome(names, norm)
    short = g.short_names
    # ---- FASTA text
    fa = []
    for name, s in zip(names, streams):
        fa.append(b">" + name.encode() + b"\n")
        i = 0
        while i < len(s):
            w = int(rng.choice([60, 60, 60, 70, 61, 1, 120]))
            fa.append(s[i:i + w].tobytes() + b"\n")
            if rng.random() < 0.02:
                fa.append(b"\n")
            i += w
    fasta = b"".join(fa)[:-1]                                # the file ends without a newline
    # ---- the regular SNPs (on the first three sequences; none can be addressed on the fourth)
    g3 = Genome(names[:3], norm[:3])
    s = make_snps(rng, g3, 2_400)
    vnames = ["1", "Un_gl000220", "contig7"]                 # "chr" is prepended to a name that does not start with 'c'
    recs = []                                                # (sort key, line)
    for i in range(len(s.pos)):
        c, ps = int(s.chrom[i]), int(s.pos[i])
        caf = "CAF=%.4f,%.4f" % (s.caf_ref[i], s.caf_alt[i])
        k = i % 7
        info = ("RS=%d;%s;COMMON=1" % (i + 1, caf), "%s;RS=%d" % (caf, i + 1), "RS=%d;PM;%s" % (i + 1, caf), "RS=%d;dbSNPBuildID=132;VP=0x05;%s;G5" % (i + 1, caf),
                "%s" % caf, "RS=%d;%s;COMMON=0" % (i + 1, caf), "RSPOS=%d;%s" % (ps, caf))[k]
        ref, alt = chr(s.ref[i]), chr(s.alt[i])
        if i % 11 == 3:
            ref, alt = ref.lower(), alt.lower()              # alleles are upper-cased (dictgen.c:634, 687)
        name = vnames[c] if i % 13 else ("chr" + vnames[c] if c < 2 else vnames[c])
        pos_txt = "%d" % ps if i % 17 else "%05d" % ps
        tail = "" if i % 19 else "\tGT:DP\t0/1:%d" % (i % 50)
        recs.append(((c, ps, 0), "%s\t%s\trs%d\t%s\t%s\t.\t.\t%s%s\n" % (name, pos_txt, i + 1, ref, alt, info, tail)))
    # ---- irregular records
    def base(c, p1):                                         # normalised base at 1-based position p1 of sequence c
        return chr(norm[c][p1 - 1])
    def other(b, k=1):
        return "ACGT"[("ACGT".index(b) + k) % 4]
    taken = set((int(c), int(q)) for c, q in zip(s.chrom, s.pos))
    def free_pos(c):
        while True:
            q = int(rng.integers(100, len(norm[c]) - 100))
            if (c, q) not in taken and all(base(c, q + d) in "ACGT" for d in range(-40, 41)):
                taken.add((c, q)); return q
    odd_recs = []
    for j in range(40):
        c = j % 3
        q = free_pos(c); b = base(c, q); caf = "CAF=0.7,0.3"
        kind = j % 10
        if kind == 0: ref, alt = b, other(b) + "," + other(b, 2)                 # multi-allelic
        elif kind == 1: ref, alt = b + base(c, q + 1), b                          # deletion
        elif kind == 2: ref, alt = b, b + "T"                                     # insertion
        elif kind == 3: ref, alt = b, b                                           # ALT equal to REF: marks the site, no k-mers
        elif kind == 4: ref, alt = b.lower(), other(b)                            # REF in lower case only
        elif kind == 5: ref, alt = b, "<DEL>"
        elif kind == 6: ref, alt = b, other(b, 2).lower()                         # ALT in lower case only
        elif kind == 7: ref, alt, caf = b, other(b), "CAF=.,0.4"                   # atof(".") = 0
        elif kind == 8: ref, alt, caf = b, other(b, 3), "CAF=9.5e-1,5e-2;CAFX=0.25,0.75"   # the last key starting with CAF wins
        else: ref, alt, caf = b, other(b, 2), "CAF=0.5,0.5;"
        odd_recs.append(((c, q, 1), "%s\t%d\tq%d\t%s\t%s\t50\tPASS\tRS=%d;%s\n" % (vnames[c], q, j, ref, alt, 900000 + j, caf)))
    for j in range(6):                                       # the same position twice: another ALT, and the very same record
        c = j % 3
        q = free_pos(c); b = base(c, q)
        odd_recs.append(((c, q, 1), "%s\t%d\td%da\t%s\t%s\t.\t.\tCAF=0.6,0.4\n" % (vnames[c], q, j, b, other(b))))
        odd_recs.append(((c, q, 2), "%s\t%d\td%db\t%s\t%s\t.\t.\tCAF=0.8,0.2\n" % (vnames[c], q, j, b, other(b, 2) if j % 2 else other(b))))
    for c in range(3):                                       # the 32-base margins (dictgen.c:675): index 31 out, 32 in, size - 32 in, size - 31 out
        L = len(norm[c])
        for q in (5, 32, 33, L - 32, L - 31, L - 30, L):
            b = base(c, q)
            if b == "N":
                odd_recs.append(((c, q, 1), "%s\t%d\tm%d\tN\tA\t.\t.\tCAF=0.5,0.5\n" % (vnames[c], q, q)))      # REF 'N' is skipped before anything is checked
            else:
                odd_recs.append(((c, q, 1), "%s\t%d\tm%d\t%s\t%s\t.\t.\tCAF=0.5,0.5\n" % (vnames[c], q, q, b, other(b))))
    for c in range(3):                                       # next to an N: the window of 32-mers runs into it (dictgen.c:760-770)
        npos = np.nonzero(norm[c] == ord("N"))[0]
        npos = npos[(npos > 200) & (npos < len(norm[c]) - 200)]
        for z in npos[:: max(1, len(npos) // 6)][:6]:
            for d in (-20, 7):
                q = int(z) + 1 + d
                b = base(c, q)
                if b != "N" and (c, q) not in taken:
                    taken.add((c, q))
                    odd_recs.append(((c, q, 1), "%s\t%d\tn%d\t%s\t%s\t.\t.\tCAF=0.9,0.1\n" % (vnames[c], q, q, b, other(b))))
    for j in range(5):
        odd_recs.append(((3, 100 + j, 1), "%s\t%d\tl%d\tA\tC\t.\t.\tCAF=0.5,0.5\n" % (short[3], 100 + j, j)))          # 64-character name: not found
        odd_recs.append(((4, 100 + j, 1), "MT\t%d\tmt%d\tA\tC\t.\t.\tCAF=0.5,0.5\n" % (100 + j, j)))                     # no such chromosome
    allr = recs + odd_recs
    allr.sort(key=lambda x: x[0])
    lines = [ln for _, ln in allr]
    blocks = [lines[i:i + 97] for i in range(0, len(lines), 97)]          # out of order: blocks of 97 lines shuffled
    order = rng.permutation(len(blocks))
    body = []
    for bi in order:
        body.extend(blocks[bi])
        if bi % 3 == 0:
            body.append("\n")
        if bi % 5 == 0:
            body.append("# a comment line in the body\n")
    vcf = VCF_HEADER + "".join(body)
    r = make_reads(rng, g3, s, 3_000, lengths=(150, 150, 101, 64, 250), err=0.01, lowq=0.30, lower_frac=0.03)
    return {"fasta": fasta, "vcf": vcf, "reads": r, "genome": g, "snps": s}


def f_repeated_records(seed=77, genome_len=40_000, n_sites=36):
    """An SNP list that holds the very same record one to five times (real dbSNP merges leave such lines behind).  `vargeno index`
    keeps every occurrence (dictgen.c:156-275: a k-mer found k times becomes one entry with an auxiliary row of k positions -- here
    the SAME position k times, or POS_AMBIGUOUS beyond 10), and `geno` turns every column of that row into a context of its own
    (qv.cc:913-933): a chunk then votes k times for one position, and each of those contexts walks the pile-up (qv.cc:1444-1494).
    A repeated record's own position is never a site (all its k-mers are ambiguous, qv.cc:637-659), so every one of them has an
    ordinary single record five bases further on, inside the same 32-mers: its counters are what the extra contexts change.
    Reads carry the ALT or the REF allele at either position (exact hits on the repeated SNP k-mers, on both strands) and
    low-quality chunks over them (neighbour hits on the same rows).
    Returns a dict like f_quirk(): fasta (bytes), vcf (str), reads (Reads)."""
    rng = np.random.default_rng(seed)
    seq = random_bases(rng, genome_len)
    fa = [b">chr1 repeated-record fixture\n"]
    for i in range(0, genome_len, 70):
        fa.append(seq[i:i + 70].tobytes() + b"\n")
    lines, reads_b, reads_q = [], [], []
    step = (genome_len - 2000) // n_sites
    for k in range(n_sites):
        q = 1000 + k * step + int(rng.integers(0, 50))        # 1-based position
        ref = chr(seq[q - 1])
        alt = "ACGT"[("ACGT".index(ref) + 1 + k % 3) % 4]
        copies = 1 + k % 5
        for z in range(copies):
            lines.append("1\t%d\trs%d\t%s\t%s\t.\t.\tCAF=0.6,0.4\n" % (q, k, ref, alt))
        ref5 = chr(seq[q + 4])
        alt5 = "ACGT"[("ACGT".index(ref5) + 2) % 4]
        lines.append("1\t%d\trs%dn\t%s\t%s\t.\t.\tCAF=0.5,0.5\n" % (q + 5, k, ref5, alt5))      # the ordinary neighbour
        if k % 7 == 3:                                        # ... and another ALT at the same position, twice
            alt2 = "ACGT"[("ACGT".index(ref) + 1 + (k + 1) % 3) % 4]
            lines += ["1\t%d\trs%db\t%s\t%s\t.\t.\tCAF=0.7,0.3\n" % (q, k, ref, alt2)] * 2
        for off in (3, 20, 45, 70, 100, 125, 140):            # the site in every chunk of a 150 bp read, and in the 22-base tail that is cut off
            st = q - 1 - off
            for allele, allele5 in ((alt, ref5), (ref, ref5), (alt, alt5), (ref, alt5)):
                b = seq[st:st + 150].copy()
                b[off] = ord(allele)
                b[off + 5] = ord(allele5)
                for rev in (False, True):
                    bb = _COMP[b[::-1]] if rev else b
                    for lowq in (False, True):
                        qq = np.full(150, ord("I"), dtype=np.uint8)
                        if lowq:
                            qq[:4] = ord("#")                 # quality character c gates chunk c (qv.cc:836)
                        reads_b.append(bb.copy()); reads_q.append(qq)
    rnd = make_reads(rng, Genome(["chr1"], [seq]), SnpSet(np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(0), np.zeros(0), np.zeros(0, np.uint8)),
                     1500, err=0.01, lowq=0.3) if False else None
    order = rng.permutation(len(reads_b))
    bases = np.concatenate([reads_b[i] for i in order])
    quals = np.concatenate([reads_q[i] for i in order])
    offsets = (np.arange(len(order) + 1, dtype=np.uint64) * np.uint64(150))
    return {"fasta": b"".join(fa), "vcf": VCF_HEADER + "".join(lines), "reads": Reads(bases, quals, offsets)}


def write_quirk(d, q):
    """The files of f_quirk() in directory d: ref.fa, snps.vcf, reads.fq."""
    with open(os.path.join(d, "ref.fa"), "wb") as f:
        f.write(q["fasta"])
    with open(os.path.join(d, "snps.vcf"), "w") as f:
        f.write(q["vcf"])
    write_fastq(os.path.join(d, "reads.fq"), q["reads"])


def genome_and_snps(seed=20261002, genome_len=40_000_000, n_snps=1_000_000, n_chroms=1, genotypes="uniform", repeats=0.0):
    """Genome + SNP list of the bench workloads (F-mid recipe scaled by length): BASELINE.json configs[1] by default; with
    genome_len = 3.1e9, n_chroms = 24, n_snps = 1e7 the hg38-scale configs[2].  Returns (genome, snps, rng): the generator is
    left where make_reads() continues from."""
    rng = np.random.default_rng(seed)
    if n_chroms == 1:
        lens, names = [genome_len], ["chr22"]
    else:
        w = np.linspace(2.0, 0.6, n_chroms)
        lens = [int(x) for x in np.floor(w / w.sum() * genome_len)]
        names = ["chr%d" % (i + 1) for i in range(n_chroms)]
    # repeats > 0: the repeat-rich stress genome (that fraction of it in families of near-identical copies, four times the microsatellites)
    g = make_genome(rng, lens, names, repeats_per_mbp=50.0, repeat_len=(200, 2000),
                    repeat_div=0.02, microsat_per_mbp=12.5 if repeats <= 0 else 50.0, exact_repeat_frac=repeats)
    s = make_snps(rng, g, n_snps, genotypes=genotypes)
    return g, s, rng


def chr22_scale(seed=20261002, genome_len=40_000_000, n_snps=1_000_000, n_reads=1_000_000, n_chroms=1, lowq=0.08):
    """BASELINE.json configs[1]: one 40 Mbp chromosome, ~1 M SNPs, 1 M x 150 bp reads (F-mid recipe)."""
    g, s, rng = genome_and_snps(seed, genome_len, n_snps, n_chroms)
    r = make_reads(rng, g, s, n_reads, lengths=(150,), err=0.005, lowq=lowq)
    return g, s, r


# ----------------------------------------------------------------------------- device-side read generator (bench.py)

class DeviceReadSource:
    """The recipe of make_reads() (150 bp, uniform start / haplotype / strand, substitution errors, the two quality
    ranges) evaluated with torch on the GPU, for bench.py: a step of the hg38-scale workload is 8 M reads and the bench
    keeps several distinct batches of its 30x stream resident, which numpy takes minutes to draw.  The two donor
    haplotypes stay on the device between calls.  Seeded (torch.Generator on the device): the same (seed, batch id)
    gives the same reads.  Plumbing only -- nothing here is on the measured path."""

    def __init__(self, genome, snps, device, seed=20261002):
        import torch

        self.torch = torch
        self.dev = device
        h0, h1, _ = haplotypes(genome, snps)
        self.h0 = torch.from_numpy(h0).to(device)
        self.h1 = torch.from_numpy(h1).to(device)
        self.G = int(len(h0))
        self.seed = int(seed)
        self.acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
        code = np.full(256, 4, dtype=np.uint8)
        comp = np.zeros(256, dtype=np.uint8)
        for i, c in enumerate(b"ACGT"):
            code[c] = i
        for a, b in zip(b"ACGTN", b"TGCAN"):
            comp[a] = b
        self.code = torch.from_numpy(code).to(device)
        self.comp = torch.from_numpy(comp).to(device)

    def batch(self, batch_id, n, length=150, err=0.005, lowq=0.08, piece=1 << 20):
        """-> (bases uint8[n*length], quals uint8[n*length], offsets int64[n+1]) on the device."""
        torch = self.torch
        gen = torch.Generator(device=self.dev)
        gen.manual_seed(self.seed * 1000003 + int(batch_id))
        bases = torch.empty(n * length, dtype=torch.uint8, device=self.dev)
        quals = torch.empty(n * length, dtype=torch.uint8, device=self.dev)
        ar = torch.arange(length, device=self.dev)
        for lo in range(0, n, piece):
            m = min(piece, n - lo)
            st = torch.randint(0, self.G - length, (m,), generator=gen, device=self.dev)
            hap = torch.randint(0, 2, (m,), generator=gen, device=self.dev).bool()
            gi = st[:, None] + ar[None, :]
            b = torch.where(hap[:, None], self.h1[gi], self.h0[gi])
            e = torch.rand((m, length), generator=gen, device=self.dev) < err
            sub = (self.code[b.long()] + torch.randint(1, 4, (m, length), generator=gen, device=self.dev, dtype=torch.uint8)) % 4
            b = torch.where(e, self.acgt[sub.long()], b)
            rv = torch.rand((m,), generator=gen, device=self.dev) < 0.5
            b = torch.where(rv[:, None], self.comp[b.flip(1).long()], b)
            bases[lo * length:(lo + m) * length] = b.reshape(-1)
            q_hi = torch.randint(ord(":"), ord("I") + 1, (m * length,), generator=gen, device=self.dev, dtype=torch.uint8)
            q_lo = torch.randint(ord("#"), ord("7") + 1, (m * length,), generator=gen, device=self.dev, dtype=torch.uint8)
            low = torch.rand((m * length,), generator=gen, device=self.dev) < lowq
            quals[lo * length:(lo + m) * length] = torch.where(low, q_lo, q_hi)
        offsets = torch.arange(0, (n + 1) * length, length, dtype=torch.int64, device=self.dev)
        return bases, quals, offsets

    def release(self):
        self.h0 = self.h1 = None


def reads_to_host(bases, quals, offsets, lo=0, hi=None):
    """Device batch (or a [lo, hi) slice of it) -> Reads on the host."""
    o = offsets.cpu().numpy().astype(np.uint64)
    hi = len(o) - 1 if hi is None else hi
    b0, b1 = int(o[lo]), int(o[hi])
    return Reads(bases[b0:b1].cpu().numpy(), quals[b0:b1].cpu().numpy(), (o[lo:hi + 1] - o[lo]).astype(np.uint64))


def f_lowcomplex(seed, genome_len=60_000, n_reads=6_000):
    """A small genome made of what makes k-mers collide with themselves and with their reverse complements: microsatellites
    of unit length 1-6 (among them A, AT, ACGT, whose runs are their own reverse complement), inverted repeats (a segment
    followed by its reverse complement: hairpins), tandem and dispersed copies (2-14 of them: auxiliary rows and, above ten,
    POS_AMBIGUOUS), between stretches of random sequence; one SNP per ~25 bases; reads of both strands, 1 % errors, 40 % gate-open
    chunks, lengths 150 / 101 / 64 / 250.  A function of the seed alone; meant to be drawn for many seeds."""
    rng = np.random.default_rng(seed)
    parts, total = [], 0
    while total < genome_len:
        kind = int(rng.integers(0, 6))
        if kind <= 1:
            seg = random_bases(rng, int(rng.integers(200, 1500)))
        elif kind == 2:
            unit = [b"A", b"AT", b"ACGT", b"AC", b"AAT", b"CG", b"AGCT", b"TTAGGG"][int(rng.integers(0, 8))]
            if rng.random() < 0.3:
                unit = bytes(random_bases(rng, int(rng.integers(1, 7))))
            seg = np.tile(np.frombuffer(unit, dtype=np.uint8), int(rng.integers(40, 400)) // len(unit) + 1)
        elif kind == 3:
            half = random_bases(rng, int(rng.integers(20, 300)))
            gap = random_bases(rng, int(rng.integers(0, 12)))
            seg = np.concatenate([half, gap, _COMP[half[::-1]]])
        elif kind == 4:
            unit = random_bases(rng, int(rng.integers(33, 400)))
            seg = np.tile(unit, int(rng.integers(2, 15)))
        else:
            if not parts:
                continue
            src = parts[int(rng.integers(0, len(parts)))]
            a = int(rng.integers(0, max(1, len(src) - 40)))
            seg = src[a:a + int(rng.integers(40, 600))].copy()
            if rng.random() < 0.5:
                seg = _COMP[seg[::-1]]
            m = rng.random(len(seg)) < (0.0 if rng.random() < 0.5 else 0.02)
            seg[m] = random_bases(rng, int(m.sum()))
        parts.append(np.ascontiguousarray(seg))
        total += len(seg)
    cat = np.concatenate(parts)
    cut = int(len(cat) * 0.6)
    g = Genome(["chr1", "chr2"], [cat[:cut].copy(), cat[cut:].copy()])
    s = make_snps(rng, g, len(cat) // 25)
    r = make_reads(rng, g, s, n_reads, lengths=(150, 150, 101, 64, 250), err=0.01, lowq=0.40)
    return g, s, r
