"""FASTQ framing on the device (vg_fastq_submit, vg_fastq_stream_begin) and on host threads inside the library
(vg_fastq_stream_begin_packed: framed + 2-bit packed on the host, r04) against the flat-batch path and, for a truncated
final record, against the reference's own output (tests/golden/ftiny.trunc.*, captured from oracle/_ref)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from conftest import BIN, GOLDEN, ROOT
from oracle import oracle as O
from vargeno_amd._lib import VgError
from vargeno_amd.api import GenoIndex

pytestmark = pytest.mark.gpu


def _counts_flat(prefix, r):
    with GenoIndex.open(prefix) as gx:
        gx.submit(r.bases, r.quals, r.offsets)
        return gx.counts(), gx.stats()


def test_device_framing_equals_flat_batches(ftiny_dir, ftiny_reads):
    prefix = os.path.join(ftiny_dir, "idx")
    text = open(os.path.join(ftiny_dir, "reads.fq"), "rb").read()
    (rc0, ac0), st0 = _counts_flat(prefix, ftiny_reads)
    with GenoIndex.open(prefix) as gx:
        # feed the file in awkward chunk sizes; the unconsumed tail of each chunk is prepended to the next
        pos, carry, total = 0, b"", 0
        for size in (70_001, 333_333, 5, 1_000_000, 10 ** 9):
            chunk = carry + text[pos:pos + size]
            pos += size
            n, used, last = gx.submit_fastq(chunk)
            total += n
            carry = chunk[used:]
            if n:
                assert chunk[last:last + 1] == b"@" and used <= len(chunk)
        assert carry == b"" and total == ftiny_reads.n
        rc, ac = gx.counts()
        st = gx.stats()
    assert np.array_equal(rc, rc0) and np.array_equal(ac, ac0)
    for k in ("reads", "reads_n", "passes", "chunks", "gate_open", "ctx", "walks", "incr"):
        assert st[k] == st0[k], k


def test_incomplete_tail_is_left_to_the_caller_and_long_lines_are_refused(ftiny_dir):
    prefix = os.path.join(ftiny_dir, "idx")
    rec = b"@r\n" + b"ACGT" * 10 + b"\n+\n" + b"I" * 40 + b"\n"
    with GenoIndex.open(prefix) as gx:
        n, used, last = gx.submit_fastq(rec * 3 + b"@r\nACGT")          # 3 records + the start of a 4th
        assert (n, used, last) == (3, 3 * len(rec), 2 * len(rec))
        n, used, _ = gx.submit_fastq(b"@r\nACGT\n+\n")                   # no complete record
        assert (n, used) == (0, 0)
        long_rec = b"@r\n" + b"A" * 1023 + b"\n+\n" + b"I" * 1023 + b"\n"   # 1024-character lines: fgets would split them
        with pytest.raises(VgError) as e:
            gx.submit_fastq(rec + long_rec)
        assert e.value.code == -6
        assert gx.stats()["reads"] == 3                                  # the refused chunk processed nothing
        short_q = b"@r\n" + b"ACGT" * 16 + b"\n+\nI\n"                  # 2 chunks, 1 quality character: the reference reads its stale buffer
        with pytest.raises(VgError) as e:
            gx.submit_fastq(short_q)
        assert e.value.code == -6
        ok_rec = b"@r\n" + b"A" * 1022 + b"\n+\n" + b"I" * 1022 + b"\n"    # 1023 characters with the newline: fine
        n, used, _ = gx.submit_fastq(ok_rec)
        assert n == 1 and used == len(ok_rec)


@pytest.mark.parametrize("host_framing,replicas,pack_threads", [("0", "1", "0"), ("1", "1", "0"), ("0", "2", "0"), ("0", "1", "3"), ("0", "2", "2")])
def test_cli_truncated_final_record_matches_the_reference(ftiny_dir, tmp_path, host_framing, replicas, pack_threads):
    """The reference's fgets() returns NULL on the missing quality line and keeps the previous record's
    buffer (qv.cc:761-763): the gate of the truncated read is the PREVIOUS read's quality string."""
    want_path = os.path.join(GOLDEN, "ftiny.trunc.out.vcf.gz")
    lines = open(os.path.join(ftiny_dir, "reads.fq"), "rb").read().split(b"\n")[:-1]
    k = int(open(os.path.join(GOLDEN, "ftiny.trunc.k")).read())
    fq = tmp_path / "reads_trunc.fq"
    fq.write_bytes(b"\n".join(lines[:4 * k + 3]))
    # (two replicas: each streams its own record-aligned half of the file; on a one-GPU box they share the device)
    # (pack_threads: 0 = the text is framed on the device; n = n host threads per replica frame + pack it inside the library)
    env = dict(os.environ, VARGENO_HOST_FASTQ=host_framing, VARGENO_CHUNK_MB="1", VARGENO_BATCH="900", VARGENO_GPUS=replicas, VARGENO_SHARE_DEVICES="1", VARGENO_PACK_THREADS=pack_threads)
    p = subprocess.run([BIN, "geno", os.path.join(ftiny_dir, "idx"), str(fq), os.path.join(ftiny_dir, "snps.vcf"), str(tmp_path / "out.vcf")],
                       env=env, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert (tmp_path / "out.vcf").read_bytes() == gzip.open(want_path, "rb").read()


@pytest.mark.parametrize("host_threads", [None, 1, 5])
def test_stream_frames_records_across_arbitrary_cuts(ftiny_dir, ftiny_reads, host_threads):
    """vg_fastq_stream_*: the file cut into chunks anywhere -- mid-line, mid-record, one byte, a whole megabyte -- gives the
    counters and event counts of the flat batch; the unfinished record is carried over inside the library and the host
    gets its only answer at the end.  host_threads None: framed on the device; n: framed + packed by n host threads."""
    prefix = os.path.join(ftiny_dir, "idx")
    text = open(os.path.join(ftiny_dir, "reads.fq"), "rb").read()
    (rc0, ac0), st0 = _counts_flat(prefix, ftiny_reads)
    rng = np.random.default_rng(3)
    for trial in range(3):
        cuts = [0]
        while cuts[-1] < len(text):
            step = int(rng.choice([1, 2, 7, 311, 4096, 65_537, 300_000, 1_000_000]))
            cuts.append(min(len(text), cuts[-1] + step))
            if len(cuts) > 400:                                       # keep the tiny steps to the head of the file
                cuts.append(len(text))
        with GenoIndex.open(prefix) as gx:
            n, used, last, refused = gx.fastq_stream((text[a:b] for a, b in zip(cuts[:-1], cuts[1:])), host_threads=host_threads)
            assert (n, used, refused) == (ftiny_reads.n, len(text), False)
            assert text[last:last + 1] == b"@" and text[last:].count(b"\n") == 4
            rc, ac = gx.counts()
            st = gx.stats()
        assert np.array_equal(rc, rc0) and np.array_equal(ac, ac0), trial
        for k in ("reads", "reads_n", "passes", "chunks", "gate_open", "ctx", "walks", "incr"):
            assert st[k] == st0[k], (trial, k)


@pytest.mark.parametrize("host_threads", [None, 4])
def test_stream_stops_at_the_first_chunk_it_cannot_frame(ftiny_dir, host_threads):
    prefix = os.path.join(ftiny_dir, "idx")
    rec = b"@r\n" + b"ACGT" * 10 + b"\n+\n" + b"I" * 40 + b"\n"
    long_rec = b"@r\n" + b"A" * 1023 + b"\n+\n" + b"I" * 1023 + b"\n"
    with GenoIndex.open(prefix) as gx:
        _stream = gx.fastq_stream
        gx.fastq_stream = lambda chunks: _stream(chunks, host_threads=host_threads)
        # an incomplete tail is simply not consumed
        n, used, last, refused = gx.fastq_stream([rec * 2 + rec[:17], rec[17:] + b"@r\nAC"])
        assert (n, used, last, refused) == (3, 3 * len(rec), 2 * len(rec), False)
        assert gx.stats()["reads"] == 3
        # a line of 1024 characters in the second chunk: the first chunk's complete records are done, nothing after them is
        gx.reset()
        head = rec * 5 + rec[:30]
        n, used, last, refused = gx.fastq_stream([head, rec[30:] + long_rec + rec * 4, rec * 100])
        assert refused and (n, used, last) == (5, 5 * len(rec), 4 * len(rec))
        assert gx.stats()["reads"] == 5
        # the handle is usable afterwards
        n, used, last, refused = gx.fastq_stream([rec * 7])
        assert (n, used, refused) == (7, 7 * len(rec), False)


def test_cli_long_line_in_the_middle_of_the_file_falls_back_to_host_framing(ftiny_dir, tmp_path):
    """A record with lines beyond fgets' 1023 characters in the middle of the file.  fgets() splits such lines, so the reference
    frames this record as TWO records made of the pieces (built here so that the pieces in read position are ACGT text -- the
    reference would abort otherwise -- and the file is back in step afterwards: 8 fgets lines).  The device refuses from that
    chunk on and the host reader takes over there; the VCF must equal the one from framing the whole file on the host."""
    lines = open(os.path.join(ftiny_dir, "reads.fq"), "rb").read().split(b"\n")[:-1]
    odd = [b"@" + b"ACGT" * 300,            # 1201 characters: fgets pieces of 1023 and 178 (+ newline); the second is read as a READ
           b"ACGT" * 20,                    # lands in separator position
           b"+",                            # lands in quality position: a 1-character quality line, the gate sees stale buffer contents
           b"ACGT" * 800]                   # 3200 characters: four pieces = id / read (1023 characters, 31 chunks) / separator / quality
    # The odd record in the first half and in the second half of the file.  Device framing with one replica: the refused chunk
    # onwards goes to the host reader.  With two replicas (each streams its own record-aligned half; they share the device on a
    # one-GPU box): in the second = last range the same hand-over happens; in the first range everything is reset and framed on
    # the host, because the range after it would be out of step with the reference.
    for at in (1500, 2500):
        k = 4 * at
        fq = tmp_path / ("reads_long_%d.fq" % at)
        fq.write_bytes(b"\n".join(lines[:k] + odd + lines[k:]) + b"\n")
        outs = []
        for host, replicas, pack in (("1", "1", "0"), ("0", "1", "0"), ("0", "2", "0"), ("0", "1", "4"), ("0", "2", "2")):
            out = tmp_path / ("out%d_%s_%s_%s.vcf" % (at, host, replicas, pack))
            env = dict(os.environ, VARGENO_HOST_FASTQ=host, VARGENO_CHUNK_MB="1", VARGENO_BATCH="900", VARGENO_READERS="3", VARGENO_VERBOSE="1",
                       VARGENO_GPUS=replicas, VARGENO_SHARE_DEVICES="1", VARGENO_PACK_THREADS=pack)
            p = subprocess.run([BIN, "geno", os.path.join(ftiny_dir, "idx"), str(fq), os.path.join(ftiny_dir, "snps.vcf"), str(out)], env=env, capture_output=True, text=True)
            assert p.returncode == 0, p.stderr
            assert "reads: %d " % (len(lines) // 4 + 2) in p.stderr, p.stderr       # the odd record counts as two
            outs.append(out.read_bytes())
        assert all(o == outs[0] for o in outs) and outs[0].count(b"\n") > 2000


def test_an_empty_stream_on_a_used_handle_reports_nothing(ftiny_dir):
    """vg_fastq_stream_begin resets the stream state on the ingest stream; with no push in between, vg_fastq_stream_end must
    still see that reset (not the previous stream's record count)."""
    prefix = os.path.join(ftiny_dir, "idx")
    text = open(os.path.join(ftiny_dir, "reads.fq"), "rb").read()
    with GenoIndex.open(prefix) as gx:
        first = gx.fastq_stream([text[:1_000_003], text[1_000_003:]])
        assert first[0] > 0 and first[1] == len(text) and not first[3]
        for _ in range(3):
            assert gx.fastq_stream([]) == (0, 0, 0, False)


@pytest.mark.parametrize("variant", ["crlf", "plus_id"])
def test_crlf_line_ends_and_named_separator_lines_against_the_reference_binary(ftiny_dir, tmp_path, variant):
    """FASTQ files as other tools write them: CRLF line ends (fgets keeps the '\\r': it becomes the last character of the read
    line, where strlen(read) - 1 cuts the '\\n' only, and of the quality line) and '+<id>' separator lines.  The REFERENCE BINARY
    genotypes the file on the GPU box's host; the product's CLI must write the same VCF through every framing: on the host,
    on the device, packed by host threads.  (Read lengths are chosen so that no '\\r' lands inside a chunk: 150 + 1 characters.)"""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "vargeno")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/vargeno not built (make -C oracle ref, build container only)")
    lines = open(os.path.join(ftiny_dir, "reads.fq"), "rb").read().split(b"\n")[:-1][:4 * 3000]
    if variant == "crlf":
        keep = [i for i in range(0, len(lines), 4) if len(lines[i + 1]) % 32 != 31]     # a '\\r' inside the last chunk would make the reference abort
        text = b"".join(b"\r\n".join(lines[i:i + 4]) + b"\r\n" for i in keep)
    else:
        text = b"".join(b"\n".join([lines[i], lines[i + 1], b"+" + lines[i][1:], lines[i + 3]]) + b"\n" for i in range(0, len(lines), 4))
    fq = tmp_path / "reads.fq"
    fq.write_bytes(text)
    want = tmp_path / "ref.vcf"
    p = subprocess.run([ref_bin, "geno", os.path.join(ftiny_dir, "idx"), str(fq), os.path.join(ftiny_dir, "snps.vcf"), str(want)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    for host, pack in (("1", "0"), ("0", "0"), ("0", "3")):
        out = tmp_path / ("out_%s_%s.vcf" % (host, pack))
        env = dict(os.environ, VARGENO_HOST_FASTQ=host, VARGENO_CHUNK_MB="1", VARGENO_PACK_THREADS=pack)
        p = subprocess.run([BIN, "geno", os.path.join(ftiny_dir, "idx"), str(fq), os.path.join(ftiny_dir, "snps.vcf"), str(out)], env=env, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        assert out.read_bytes() == want.read_bytes(), (variant, host, pack)
    assert want.read_bytes().count(b"\n") > 100


def test_read_store_filled_before_the_index_is_open(ftiny_dir, ftiny_reads):
    """vg_read_store_*: batches packed and parked in device memory BEFORE the handle exists (what the command line does while
    vg_index_open runs), then run through the handle -- twice: the counters of the flat ASCII batch, then twice them (saturating at
    255 like every counter); a full store refuses the batch and keeps what it has; a store of another device is refused."""
    from vargeno_amd.api import HostPacker, ReadStore

    prefix = os.path.join(ftiny_dir, "idx")
    text = open(os.path.join(ftiny_dir, "reads.fq"), "rb").read()
    (rc0, ac0), st0 = _counts_flat(prefix, ftiny_reads)
    pk = HostPacker(2)
    pk.begin()
    store = ReadStore(0, 64 << 20)
    small = ReadStore(0, 4096)
    total = 0
    for a in range(0, len(text), 555_555):
        k, m, o, bad = pk.push(text[a:a + 555_555])
        total += len(m)
        if len(m):
            store.push(k, m, o)
            with pytest.raises(VgError) as e:
                small.push(k, m, o)
            assert e.value.code == -3
    assert total == ftiny_reads.n and store.reads == total and small.reads == 0 and store.bytes_used > 0
    pk.close()
    small.close()
    with GenoIndex.open(prefix) as gx:
        gx.set_stats(True)
        gx.submit_store(store)
        rc, ac = gx.counts()
        assert np.array_equal(rc, rc0) and np.array_equal(ac, ac0)
        st = gx.stats()
        for key in ("reads", "reads_n", "reads_invalid", "passes", "passes_ok", "chunks", "gate_open", "ref_query", "snp_query", "ctx", "walks", "incr"):
            assert st[key] == st0[key], key
        gx.set_stats(False)
        gx.submit_store(store)
        rc2, ac2 = gx.counts()
        assert np.array_equal(rc2, np.minimum(2 * rc0.astype(np.int64), 255)) and np.array_equal(ac2, np.minimum(2 * ac0.astype(np.int64), 255))
        # the rules of vg_reads_submit_packed hold at the store's door
        with pytest.raises(VgError) as e:
            store.push(np.zeros(32, np.uint64), np.zeros(1, np.uint64), np.array([0, 32], np.uint64))
        assert e.value.code == -6
    store.close()


@pytest.mark.parametrize("force_generic", ["0", "1"])
def test_packed_batches_equal_flat_batches_through_every_tier(ftiny_dir, ftiny_reads, monkeypatch, force_generic):
    """vg_reads_submit_packed with batches framed + packed by the library's host packer (vg_packer_*), in odd batch sizes: site
    counters and event counters of the flat ASCII batch.  With VG_FORCE_GENERIC=1 the lane machine alone reads the packed form."""
    from vargeno_amd.api import HostPacker

    prefix = os.path.join(ftiny_dir, "idx")
    text = open(os.path.join(ftiny_dir, "reads.fq"), "rb").read()
    (rc0, ac0), st0 = _counts_flat(prefix, ftiny_reads)
    monkeypatch.setenv("VG_FORCE_GENERIC", force_generic)
    pk = HostPacker(3)
    with GenoIndex.open(prefix) as gx:
        for stats in (True, False):
            gx.reset()
            gx.set_stats(stats)
            pk.begin()
            total = 0
            for a in range(0, len(text), 777_777):
                k, m, o, bad = pk.push(text[a:a + 777_777])
                total += len(m)
                if len(m):
                    gx.submit_packed(k, m, o)
            assert total == ftiny_reads.n and pk.end()[1] == len(text)
            rc, ac = gx.counts()
            assert np.array_equal(rc, rc0) and np.array_equal(ac, ac0), stats
            if stats:
                st = gx.stats()
                for key in ("reads", "reads_n", "reads_invalid", "passes", "passes_ok", "chunks", "gate_open", "ref_query", "snp_query", "ctx", "walks", "incr"):
                    assert st[key] == st0[key], key
        # a packed read of more than 31 chunks cannot come from a FASTQ line the reference reads: refused
        with pytest.raises(VgError) as e:
            gx.submit_packed(np.zeros(32, np.uint64), np.zeros(1, np.uint64), np.array([0, 32], np.uint64))
        assert e.value.code == -6
    pk.close()


@pytest.mark.parametrize("replicas", ["1", "2"])
def test_cli_read_store_that_fills_up_in_the_middle_of_the_file(ftiny_dir, tmp_path, replicas):
    """The command line packs the file ahead of the index into a read store (device memory); a 30x file does not fit the store, so
    the pre-packer must stop at the last chunk the store took and the rest of the range must be framed after the open, from exactly
    that record on -- by host threads or on the device, whichever the measurement picks.  F-tiny's reads five times over (5.2 MB, five
    1 MiB chunks; with two replicas, two and a half each), the store sized exactly (VARGENO_PREPACK_BYTES): for nothing, for one
    chunk, for two chunks, for everything.  Every run must write the VCF of the run that packs nothing ahead (VARGENO_PREPACK=0),
    and the middle sizes must really have stopped in the middle (the verbose line says how many reads were packed ahead)."""
    import re

    from vargeno_amd.api import HostPacker

    text = open(os.path.join(ftiny_dir, "reads.fq"), "rb").read() * 5
    fq = tmp_path / "reads5.fq"
    fq.write_bytes(text)
    n_total = text.count(b"\n") // 4
    # what the first 1 MiB chunk of a range takes in the store: three arrays, each rounded up to 256 bytes
    pk = HostPacker(1)
    pk.begin()
    k, m, o, _ = pk.push(text[:1 << 20])
    pk.close()
    up = lambda b: (b + 255) // 256 * 256
    one_chunk = up((len(k) + 2) * 8) + up(len(m) * 8) + up((len(m) + 1) * 8)

    def run(extra):
        out = tmp_path / ("out_%s.vcf" % "_".join("%s%s" % kv for kv in sorted(extra.items())))
        env = dict(os.environ, VARGENO_CHUNK_MB="1", VARGENO_PACK_THREADS="3", VARGENO_GPUS=replicas, VARGENO_SHARE_DEVICES="1", VARGENO_VERBOSE="1", **extra)
        p = subprocess.run([BIN, "geno", os.path.join(ftiny_dir, "idx"), str(fq), os.path.join(ftiny_dir, "snps.vcf"), str(out)], env=env, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        ahead = [int(x) for x in re.findall(r"ingest, replica \d+: (\d+) reads packed ahead", p.stderr)]
        return out.read_bytes(), ahead, p.stderr

    want, ahead, _ = run({"VARGENO_PREPACK": "0"})
    assert ahead == [] and want.count(b"\n") > 100
    # (reads.fq is just under 1 MiB: `one_chunk` is what ~4 000 reads take, a 1 MiB chunk holds a few more; a replica's range is
    # five such chunks, or two and a half with two replicas)
    for size, expect in ((4096, "none"), (one_chunk * 13 // 10, "some"), (one_chunk * 23 // 10, "some"), (64 << 20, "all")):
        got, ahead, err = run({"VARGENO_PREPACK_BYTES": str(size)})
        assert got == want, (size, err)
        assert len(ahead) == int(replicas), err
        if expect == "none":
            assert sum(ahead) == 0, err
        elif expect == "all":
            assert sum(ahead) == n_total, err
        else:
            assert 0 < sum(ahead) < n_total, (size, ahead, err)
            assert "when the store was full" in err and "rest of the range" in err, err


@pytest.mark.parametrize("replicas", ["1", "2"])
def test_cli_reads_a_fastq_that_is_not_a_regular_file(ftiny_dir, tmp_path, replicas):
    """`vargeno geno idx <(zcat reads.fq.gz) ...` -- the reference fopen()s whatever path it is given and fgets its way through it
    (qv.cc:2182, 760-763), so a FIFO, /dev/stdin and bash's process substitution all work there.  The drop-in's file routes pread
    ranges from many threads and re-open the path for the host reader; a path that is not a regular file takes the once-only route
    instead (PipeIngest: one descriptor, one reader thread, the host packer, batches round robin over the replicas -- into their
    read stores while the index opens, then straight into the read loop).  F-tiny through a FIFO fed by a writer thread, through
    /dev/stdin and through <(cat ...), with a read store sized for everything, for about one chunk and for nothing; and the
    truncated-final-record file (the reference's stale line buffers, qv.cc:761-763): the golden VCFs, byte for byte."""
    import re
    import threading

    idx, snps = os.path.join(ftiny_dir, "idx"), os.path.join(ftiny_dir, "snps.vcf")
    whole = open(os.path.join(ftiny_dir, "reads.fq"), "rb").read()
    lines = whole.split(b"\n")[:-1]
    k = int(open(os.path.join(GOLDEN, "ftiny.trunc.k")).read())
    trunc = b"\n".join(lines[:4 * k + 3])
    n_whole = len(lines) // 4
    golden = {"whole": gzip.open(os.path.join(GOLDEN, "ftiny.out.vcf.gz"), "rb").read(), "trunc": gzip.open(os.path.join(GOLDEN, "ftiny.trunc.out.vcf.gz"), "rb").read()}
    base_env = dict(os.environ, VARGENO_CHUNK_MB="1", VARGENO_PACK_THREADS="2", VARGENO_GPUS=replicas, VARGENO_SHARE_DEVICES="1", VARGENO_VERBOSE="1")
    n = [0]

    def through_fifo(data, extra):
        n[0] += 1
        fifo, out = str(tmp_path / ("in%d.fifo" % n[0])), tmp_path / ("out%d.vcf" % n[0])
        os.mkfifo(fifo)

        def feed():
            with open(fifo, "wb", buffering=0) as w:
                for a in range(0, len(data), 300_000):
                    w.write(data[a:a + 300_000])
        t = threading.Thread(target=feed)
        t.start()
        p = subprocess.run([BIN, "geno", idx, fifo, snps, str(out)], env=dict(base_env, **extra), capture_output=True, text=True, timeout=300)
        t.join()
        assert p.returncode == 0, p.stderr
        return out.read_bytes(), p.stderr

    for size in ("67108864", "300000", "4096"):                         # the read store: everything fits / about one chunk / nothing
        got, err = through_fifo(whole, {"VARGENO_PREPACK_BYTES": size})
        assert got == golden["whole"], (size, err)
        m = re.search(r"not a regular file: one descriptor read once, (\d+) reads framed \+ packed by \d+ host threads \((\d+) into the read stores while the index opened, (\d+) straight", err)
        assert m and int(m.group(1)) == n_whole and int(m.group(2)) + int(m.group(3)) == n_whole, err
        if size == "4096":
            assert int(m.group(2)) == 0, err
    got, err = through_fifo(trunc, {})
    assert got == golden["trunc"], err
    got, err = through_fifo(whole, {"VARGENO_PREPACK": "0"})
    assert got == golden["whole"], err
    # /dev/stdin
    out = tmp_path / "out_stdin.vcf"
    p = subprocess.run([BIN, "geno", idx, "/dev/stdin", snps, str(out)], env=base_env, input=trunc, capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert out.read_bytes() == golden["trunc"]
    # bash process substitution
    fq = tmp_path / "reads.fq"
    fq.write_bytes(whole)
    out = tmp_path / "out_subst.vcf"
    p = subprocess.run(["bash", "-c", "%s geno %s <(cat %s) %s %s" % (str(BIN), idx, fq, snps, out)], env=base_env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert out.read_bytes() == golden["whole"]


@pytest.mark.parametrize("copiers", ["4", "0"])
def test_cli_reads_a_fifo_whose_writer_lends_its_pages(ftiny_dir, tmp_path, copiers):
    """A producer may hand a pipe its own pages instead of copying into it (vmsplice: what bench.py's `job_stream` feed does), and the
    command line's reader moves a pipe's pages on to its copier threads (splice) instead of read()ing them: F-tiny's FASTQ lent to a
    FIFO in one piece, small chunks, with the copier threads and with the plain read() loop -- the golden VCF, byte for byte, and every
    record framed by the packer."""
    import errno
    import importlib.util
    import mmap
    import re
    import sys
    import time

    spec = importlib.util.spec_from_file_location("vg_bench_mod3", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    idx, snps = os.path.join(ftiny_dir, "idx"), os.path.join(ftiny_dir, "snps.vcf")
    whole = open(os.path.join(ftiny_dir, "reads.fq"), "rb").read()
    n_whole = whole.count(b"\n") // 4
    golden = gzip.open(os.path.join(GOLDEN, "ftiny.out.vcf.gz"), "rb").read()
    m = mmap.mmap(-1, len(whole))
    m[:] = whole
    fifo, out = str(tmp_path / "lent.fifo"), tmp_path / "lent.vcf"
    os.mkfifo(fifo)
    env = dict(os.environ, VARGENO_CHUNK_MB="1", VARGENO_PACK_THREADS="2", VARGENO_VERBOSE="1", VARGENO_PIPE_COPIERS=copiers)
    p = subprocess.Popen([BIN, "geno", idx, fifo, snps, str(out)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    fd, t0 = None, time.time()
    while fd is None:                                                   # (a FIFO opens for writing once its reader is there)
        try:
            fd = os.open(fifo, os.O_WRONLY | os.O_NONBLOCK)
        except OSError as e:
            assert e.errno == errno.ENXIO and p.poll() is None and time.time() - t0 < 120, (e, p.poll())
            time.sleep(0.01)
    import fcntl

    fcntl.fcntl(fd, fcntl.F_SETFL, fcntl.fcntl(fd, fcntl.F_GETFL) & ~os.O_NONBLOCK)
    feed = bench.FifoFeed(fd, lend=True)
    feed.pipe_bytes = 1 << 16            # (the copied tail is 2 x this: F-tiny's FASTQ is one megabyte, and this buffer is never refilled)
    feed.write_all(memoryview(m))
    os.close(fd)
    so, se = p.communicate(timeout=300)
    assert p.returncode == 0, se
    assert out.read_bytes() == golden, se
    mt = re.search(r"not a regular file: one descriptor read once, (\d+) reads framed .* dealt to (\d+) copier threads", se)
    assert mt and int(mt.group(1)) == n_whole and int(mt.group(2)) == int(copiers), se
    if feed.refusal is None:
        assert feed.lent_bytes > len(whole) // 2, (feed.lent_bytes, feed.copied_bytes)


@pytest.mark.parametrize("replicas", ["1", "2"])
def test_cli_long_line_in_the_middle_of_a_fifo_goes_on_through_the_host_reader(ftiny_dir, tmp_path, replicas):
    """The once-only route's hand-over on the device path: a record with lines beyond fgets' 1023 characters (see the file test above)
    in the middle of a stream that arrives through a FIFO.  The packer refuses from that chunk on; the host reader gets what the
    reader thread (and its copier threads) had already taken from the pipe, then the descriptor.  The VCF must equal the one from
    framing the same bytes, as a file, on the host -- with the copier threads and with the plain read() loop."""
    import threading

    lines = open(os.path.join(ftiny_dir, "reads.fq"), "rb").read().split(b"\n")[:-1]
    odd = [b"@" + b"ACGT" * 300, b"ACGT" * 20, b"+", b"ACGT" * 800]
    k = 4 * 1500
    data = b"\n".join(lines[:k] + odd + lines[k:]) + b"\n"
    fq = tmp_path / "reads_long.fq"
    fq.write_bytes(data)
    idx, snps = os.path.join(ftiny_dir, "idx"), os.path.join(ftiny_dir, "snps.vcf")
    base_env = dict(os.environ, VARGENO_CHUNK_MB="1", VARGENO_BATCH="900", VARGENO_VERBOSE="1", VARGENO_GPUS=replicas, VARGENO_SHARE_DEVICES="1", VARGENO_PACK_THREADS="2")
    want_out = tmp_path / "want.vcf"
    p = subprocess.run([BIN, "geno", idx, str(fq), snps, str(want_out)], env=dict(base_env, VARGENO_HOST_FASTQ="1", VARGENO_GPUS="1"), capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    want = want_out.read_bytes()
    assert want.count(b"\n") > 2000
    for copiers in ("4", "0"):
        fifo, out = str(tmp_path / ("long_%s.fifo" % copiers)), tmp_path / ("long_%s.vcf" % copiers)
        os.mkfifo(fifo)

        def feed():
            with open(fifo, "wb", buffering=0) as w:
                for a in range(0, len(data), 70_000):
                    w.write(data[a:a + 70_000])
        t = threading.Thread(target=feed)
        t.start()
        p = subprocess.run([BIN, "geno", idx, fifo, snps, str(out)], env=dict(base_env, VARGENO_PIPE_COPIERS=copiers), capture_output=True, text=True, timeout=300)
        t.join()
        assert p.returncode == 0, p.stderr
        assert "the stream framing refused a chunk" in p.stderr and "reads: %d " % (len(lines) // 4 + 2) in p.stderr, p.stderr
        assert out.read_bytes() == want, (copiers, p.stderr)
