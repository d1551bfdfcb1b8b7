#!/bin/bash
# Round-2 closing run on one box lease: the profile of the default bench (run_prof_r02.sh) and the closing validation
# (run_r02_final.sh: whole GPU test-suite, smoke(), default + chr22 bench, CLI end to end at hg38 scale).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash profiles/run_prof_r02.sh r02
cd $R
bash profiles/run_r02_final.sh
