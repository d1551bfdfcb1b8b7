#!/bin/bash
# Round 5: the command line with a read store that fills up in the middle of the file (tests/test_gpu_fastq.py), and the store's own test.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_store
mkdir -p $OUT
cd $R
( time timeout 600 python3 -m pytest tests/test_gpu_fastq.py -m gpu -q -x -k "store" ) > $OUT/pytest.txt 2>&1
tail -30 $OUT/pytest.txt | cut -c1-300
