#!/bin/bash
# unprofiled step rate of builds of the library, alternating, three runs each: bash tools/lib_sweep.sh [B=256] "base <variant.so> base ..."
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; cd $R; B=$1; shift
what=$([ "$B" = 256 ] && echo train27 || echo small)
for tag in $1; do
  if [ $tag = base ]; then unset KASF_LIB; else export KASF_LIB=$R/kasportsformer_amd/$tag; fi
  r=""
  for k in 1 2 3; do
    v=$(python tools/bench_configs.py $what 2>/dev/null | grep '^{' | head -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['clips_per_s']))")
    r="$r $v"
  done
  echo "$tag  clips/s:$r"
done
unset KASF_LIB
