#!/bin/bash
# the whole GPU suite (writes the parity reports the bench line quotes), then the one-pass profile refresh
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -s > $O/pytest_full.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_full.log
bash tools/refresh_profiles_r4.sh > $O/refresh.log 2>&1; echo "refresh rc=$?"; tail -30 $O/refresh.log
