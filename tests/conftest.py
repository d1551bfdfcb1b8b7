import gzip
import os
import shutil
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_context_first():
    """On a GPU box create torch's HIP context before the library's first big allocations: torch initialised late,
    after several 35 GB index open/close cycles in the same process, has been seen to report "No HIP GPUs are
    available" (torch only does plumbing in these tests: device buffers for vg_reads_process_device)."""
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda:0")
    except Exception:
        pass
    yield


def _gunzip(src, dst):
    with gzip.open(src, "rb") as f, open(dst, "wb") as g:
        shutil.copyfileobj(f, g)


@pytest.fixture(scope="session")
def ftiny_dir(tmp_path_factory):
    """The committed F-tiny fixture as files: inputs + the index the REFERENCE wrote for them
    (bit-vector files re-created sparse from the committed set-bit lists)."""
    from vargeno_amd import index_io

    d = str(tmp_path_factory.mktemp("ftiny"))
    for name in ("ref.fa", "snps.vcf", "reads.fq", "idx.ref.dict", "idx.snp.dict"):
        _gunzip(os.path.join(GOLDEN, "ftiny.%s.gz" % name), os.path.join(d, name))
    shutil.copy(os.path.join(GOLDEN, "ftiny.idx.chrlens"), os.path.join(d, "idx.chrlens"))
    z = np.load(os.path.join(GOLDEN, "ftiny.bf.npz"))
    index_io.write_bf_sparse(os.path.join(d, "idx.ref.bf"), int(z["ref_bits"]), z["ref_set"])
    index_io.write_bf_sparse(os.path.join(d, "idx.snp.bf"), int(z["snp_bits"]), z["snp_set"])
    _gunzip(os.path.join(GOLDEN, "ftiny.out.vcf.gz"), os.path.join(d, "golden.out.vcf"))
    return d


@pytest.fixture(scope="session")
def ftiny_reads():
    from vargeno_amd import synth

    return synth.f_tiny()[2]


def read_sha256_list(name):
    out = {}
    with open(os.path.join(GOLDEN, name + ".sha256")) as f:
        for line in f:
            h, fn = line.split()
            out[fn] = h
    return out
