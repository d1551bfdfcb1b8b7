#!/bin/bash
# Round 5, lease j: pack tiles sized by the read length, asynchronous hand-over of the pre-packed batches, VCF text read beside the open:
# FASTQ / CLI / parity tests, then the job leg (200 M reads) and the read-length legs on the open index.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_j
mkdir -p $OUT
cd $R
( time timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fastq.py tests/test_gpu_cli.py -m gpu -q -x -k "not hg38" ) > $OUT/pytest.txt 2>&1
tail -6 $OUT/pytest.txt
bash profiles/run_r05_stage.sh j --secondary len101,len250 --cpu-reference no --no-gather-probe
