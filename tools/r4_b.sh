#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4b; mkdir -p $O
cd $R
python tools/dp_probe.py 32 24 > $O/dp_probe32.jsonl 2> $O/dp_probe32.err; echo "dp_probe rc=$?"; cat $O/dp_probe32.jsonl | cut -c1-400
python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
