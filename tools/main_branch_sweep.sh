#!/bin/bash
# which branch of a layer runs on the caller's stream (KASF_MAIN_BRANCH_FWD / _BWD: 0 attention, 1 graph, 2 bone) against the step rate, same box:
#   bash tools/main_branch_sweep.sh [B=256] "0 0" "1 2" ...
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; cd $R; B=$1; shift
for p in "$@"; do
  set -- $p
  r=""
  for k in 1 2 3; do
    if [ "$B" = 256 ]; then
      v=$(KASF_MAIN_BRANCH_FWD=$1 KASF_MAIN_BRANCH_BWD=$2 python tools/bench_configs.py train27 2>/dev/null | grep '^{' | head -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['clips_per_s']))")
    else
      v=$(KASF_MAIN_BRANCH_FWD=$1 KASF_MAIN_BRANCH_BWD=$2 python tools/bench_configs.py small 2>/dev/null | grep '^{' | head -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['clips_per_s']))")
    fi
    r="$r $v"
  done
  echo "main branch fwd/bwd $p  clips/s:$r"
done
