#!/bin/bash
# grid widths of the persistent launches (KASF_NARROW_PCTS = fwd,bwd,dgrad,linear,attn_fwd,attn_bwd,wgrad) at the per-rank batches of configs[2] (B = 32 / 64, detector-confidence
# input, data-parallel path), same box:   bash tools/width_sweep_small.sh "50,50,50,50,100,100,100" "33,33,33,33,100,100,100" ...
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; cd $R
for p in "$@"; do
  export KASF_NARROW_PCTS=$p
  r=$(python tools/bench_configs.py small 2>/dev/null | grep '^{' | python3 -c "
import json,sys
print(' | '.join(f\"{d['config'][6:40]}: {d['clips_per_s']:.0f}\" for d in map(json.loads, sys.stdin)))")
  echo "pcts $p :: $r"
done
