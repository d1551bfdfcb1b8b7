"""End to end through the drop-in command line on a GPU: `vargeno index` + `vargeno geno` must
write the very bytes the reference wrote (tests/golden/*.out.vcf.gz, captured from oracle/_ref)."""
import gzip
import os
import subprocess

import pytest

from conftest import BIN, GOLDEN, ROOT
from vargeno_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,gen", [("ftiny", synth.f_tiny), ("fsmall", synth.f_small), ("fquirk", synth.f_quirk), ("frepeated", synth.f_repeated_records)])
def test_cli_vcf_is_byte_identical_to_the_reference(name, gen, tmp_path):
    d = str(tmp_path)
    if name in ("fquirk", "frepeated"):                                   # irregular FASTA / SNP-list inputs (synth.f_quirk); records held up to five times
        synth.write_quirk(d, gen())
    else:
        g, s, r = gen()
        synth.write_fasta(os.path.join(d, "ref.fa"), g)
        synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
        synth.write_fastq(os.path.join(d, "reads.fq"), r)
    env = dict(os.environ, VARGENO_NO_LITE="1", VARGENO_BATCH="7000")     # several batches
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=env, stdout=subprocess.DEVNULL)
    p = subprocess.run([BIN, "geno", "idx", "reads.fq", "snps.vcf", "out.vcf"], cwd=d, env=env, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert p.stderr.startswith("Initializing...\nProcessing...\n") and p.stdout.startswith("Time: ")
    want = gzip.open(os.path.join(GOLDEN, name + ".out.vcf.gz"), "rb").read()
    got = open(os.path.join(d, "out.vcf"), "rb").read()
    assert got == want


def test_cli_without_index_fails_cleanly(tmp_path):
    p = subprocess.run([BIN, "geno", "nope", "reads.fq", "snps.vcf", "out.vcf"], cwd=str(tmp_path), capture_output=True, text=True)
    assert p.returncode == 1


REF_BIN = os.path.join(ROOT, "oracle", "_ref", "vargeno")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/vargeno (the reference built by oracle/Makefile in the build container) is not there")
def test_cli_matches_the_reference_binary_at_scale(tmp_path):
    """The reference ITSELF (oracle/_ref/vargeno: its own sources compiled by oracle/Makefile, carried to the GPU box as a
    binary) and the product, both run here on the same chr22-scale index files and the same 300 000 reads: the two VCFs
    must be the same bytes.  (The reference needs ~20 GB of host memory and ~15 s for its jump table.)"""
    g, s, r = synth.chr22_scale(n_reads=300_000)
    d = str(tmp_path)
    synth.write_fasta(os.path.join(d, "ref.fa"), g)
    synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
    synth.write_fastq(os.path.join(d, "reads.fq"), r)
    env = dict(os.environ, VARGENO_NO_LITE="1")
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=env, stdout=subprocess.DEVNULL)
    p = subprocess.run([BIN, "geno", "idx", "reads.fq", "snps.vcf", "ours.vcf"], cwd=d, env=env, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    q = subprocess.run([REF_BIN, "geno", "idx", "reads.fq", "snps.vcf", "ref.vcf"], cwd=d, capture_output=True, text=True, timeout=900)
    assert q.returncode == 0, q.stderr
    ours = open(os.path.join(d, "ours.vcf"), "rb").read()
    ref = open(os.path.join(d, "ref.vcf"), "rb").read()
    assert ours.count(b"\n") > 100_000
    assert ours == ref
    # r06: the same job under a 10 GB device budget (tables of the index's own size: ~7.6 GB instead of 91) -- the same bytes; and with a
    # direct table forced down to 2^16 buckets, whose buckets then hold ~1 100 entries where the record's count field has 8 bits: the loader
    # must notice, give the merged view up (the look-up falls back to the dictionaries' own tables) and say so -- the same bytes again
    for extra, must_say in (({"VARGENO_MAX_DEVICE_GB": "10"}, "direct table of 2^2"), ({"VG_DX_BITS": "16"}, "not kept: a bucket of more than 255 entries")):
        p = subprocess.run([BIN, "geno", "idx", "reads.fq", "snps.vcf", "ours2.vcf"], cwd=d, env=dict(env, VARGENO_VERBOSE="1", **extra), capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        assert must_say in p.stderr, p.stderr
        assert open(os.path.join(d, "ours2.vcf"), "rb").read() == ref, extra


def _hg38_cli_job(tmp_dir):
    """BASELINE.json configs[2] shape end to end: the reference binary and the product's `vargeno geno`, both on the hg38-scale
    index files (shared with bench.py / test_gpu_fullsize.py through VG_BENCH_DIR) and the same 200 000 reads.  The reference
    spends ~4 minutes of its run loading the 43 GB dictionary field by field, on one thread: it is started here and left
    running beside the tests that follow; tests/test_gpu_zz_hg38_reference.py, the last module of the suite, compares."""
    import torch

    d = os.environ.get("VG_BENCH_DIR", "/tmp/vg_bench") + "/g3100000000_s10000000_c24"
    g, s, _ = synth.genome_and_snps(genome_len=3_100_000_000, n_snps=10_000_000, n_chroms=24)
    if not os.path.exists(d + "/idx.done"):
        os.makedirs(d, exist_ok=True)
        synth.write_fasta(d + "/ref.fa", g)
        synth.write_vcf(d + "/snps.vcf", g, s)
        subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
        open(d + "/idx.done", "w").close()
    src = synth.DeviceReadSource(g, s, torch.device("cuda", 0))
    del g, s
    r = synth.reads_to_host(*src.batch(777, 200_000))
    src.release()
    del src
    torch.cuda.empty_cache()
    fq = os.path.join(tmp_dir, "reads.fq")
    synth.write_fastq(fq, r)
    ref = subprocess.Popen([REF_BIN, "geno", "idx", fq, "snps.vcf", os.path.join(tmp_dir, "ref.vcf")], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    job = {"ref": ref, "ref_vcf": os.path.join(tmp_dir, "ref.vcf"), "ours_vcf": os.path.join(tmp_dir, "ours.vcf"), "ours_rc": None, "ours_err": ""}
    try:
        p = subprocess.run([BIN, "geno", "idx", fq, "snps.vcf", job["ours_vcf"]], cwd=d, capture_output=True, text=True)
        job["ours_rc"], job["ours_err"] = p.returncode, p.stderr
    except BaseException:
        ref.kill()
        raise
    return job


def finish_hg38_cli_job(job):
    try:
        assert job["ours_rc"] == 0, job["ours_err"]
        assert job["ref"].wait(timeout=900) == 0
    finally:
        if job["ref"].poll() is None:
            job["ref"].kill()
    ours = open(job["ours_vcf"], "rb").read()
    assert ours.count(b"\n") > 50_000
    assert ours == open(job["ref_vcf"], "rb").read()


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/vargeno (the reference built by oracle/Makefile in the build container) is not there")
def test_cli_at_hg38_scale_runs_and_the_reference_is_started(tmp_path_factory):
    """First half of the hg38-scale end-to-end check (the second is test_gpu_zz_hg38_reference.py): the product's `vargeno geno`
    runs to completion here; the reference binary, which needs minutes, keeps running beside the following tests."""
    import conftest

    job = _hg38_cli_job(str(tmp_path_factory.mktemp("hg38cli")))
    conftest.BACKGROUND["hg38_cli"] = job
    assert job["ours_rc"] == 0, job["ours_err"]
    assert open(job["ours_vcf"], "rb").read().count(b"\n") > 50_000
