#!/bin/bash
# Round 5, lease w: the driver's command on the shipped library (all secondary legs, the job leg), watched.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash profiles/run_r05_stage.sh w
