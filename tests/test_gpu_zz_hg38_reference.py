"""Last module of the GPU suite: the hg38-scale end-to-end comparison whose reference process tests/test_gpu_cli.py started
(the reference binary needs ~4 minutes, nearly all of it start-up on one thread; run beside the other tests it costs the
suite nothing)."""
import os

import pytest

import conftest
from test_gpu_cli import REF_BIN, _hg38_cli_job, finish_hg38_cli_job

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/vargeno (the reference built by oracle/Makefile in the build container) is not there")
def test_cli_matches_the_reference_binary_at_hg38_scale(tmp_path_factory):
    """The two VCFs -- the reference binary's and the product's, same hg38-scale index files, same 200 000 reads -- must be the
    same bytes.  (Run on its own, this test does both halves itself.)"""
    job = conftest.BACKGROUND.pop("hg38_cli", None)
    if job is None:
        job = _hg38_cli_job(str(tmp_path_factory.mktemp("hg38cli")))
    finish_hg38_cli_job(job)
