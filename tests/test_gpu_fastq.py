"""FASTQ framing on the device (vg_fastq_submit) against the flat-batch path and, for a truncated final record,
against the reference's own output (tests/golden/ftiny.trunc.*, captured from oracle/_ref)."""
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


@pytest.mark.parametrize("host_framing", ["0", "1"])
def test_cli_truncated_final_record_matches_the_reference(ftiny_dir, tmp_path, host_framing):
    """The reference's fgets() returns NULL on the missing quality line and keeps the previous record's
    buffer (qv.cc:761-763): the gate of the truncated read is the PREVIOUS read's quality string."""
    want_path = os.path.join(GOLDEN, "ftiny.trunc.out.vcf.gz")
    lines = open(os.path.join(ftiny_dir, "reads.fq"), "rb").read().split(b"\n")[:-1]
    k = int(open(os.path.join(GOLDEN, "ftiny.trunc.k")).read())
    fq = tmp_path / "reads_trunc.fq"
    fq.write_bytes(b"\n".join(lines[:4 * k + 3]))
    env = dict(os.environ, VARGENO_HOST_FASTQ=host_framing, VARGENO_CHUNK_MB="1", VARGENO_BATCH="900")
    p = subprocess.run([BIN, "geno", os.path.join(ftiny_dir, "idx"), str(fq), os.path.join(ftiny_dir, "snps.vcf"), str(tmp_path / "out.vcf")],
                       env=env, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert (tmp_path / "out.vcf").read_bytes() == gzip.open(want_path, "rb").read()
