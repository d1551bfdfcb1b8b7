#!/bin/bash
# Round 5, last lease: the driver's command with the chr22 and lowq50 legs only (no reference binary, no ingest / job legs), after the
# counters' fetch joined the warm-up: the in-bench chr22 leg.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash profiles/run_r05_stage.sh zz --secondary chr22,lowq50 --cpu-reference no --no-ingest --job-reads 0
