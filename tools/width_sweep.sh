#!/bin/bash
# grid widths of the persistent launches (KASF_NARROW_PCTS: mlp fwd, mlp bwd, dgrad, linear, attn fwd, attn bwd, wgrad jobs; percent of the full grid) against the step rate, same box:
#   bash tools/width_sweep.sh [B=256] "50,50,50,50,100,100,100" "66,50,..." ...
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; cd $R; B=$1; shift
what=$([ "$B" = 256 ] && echo train27 || echo small)
for p in "$@"; do
  r=""
  for k in 1 2 3; do
    v=$(KASF_NARROW_PCTS=$p KASF_NARROW_BELOW=100000000 python tools/bench_configs.py $what 2>/dev/null | grep '^{' | head -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['clips_per_s']))")
    r="$r $v"
  done
  echo "pcts $p  clips/s:$r"
done
