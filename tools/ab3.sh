#!/bin/bash
# same-box A/B of builds of the library INSIDE the three-stream step: bash tools/ab3.sh "<variant.so> ..." [kernel regex] [B]
# per build: three-stream kernel trace of three training steps (in-step averages of the kernels matching the regex), then the unprofiled step rate three times
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; VS=$1; RX=${2:-k_mlp}; B=${3:-256}; O=$R/gpurun_out/ab3; mkdir -p $O
for tag in base $VS; do
  if [ $tag = base ]; then unset KASF_LIB; else export KASF_LIB=$R/kasportsformer_amd/$tag; fi
  rm -rf $O/p_$tag
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p_$tag -o t --output-format csv -- python3 $R/tools/train_once.py 27 $B > $O/p_$tag.log 2>&1)
  echo "== $tag"
  python3 - $O/p_$tag/t_kernel_stats.csv "$RX" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(f"   {r['Name'][:58]:58s} n/step {int(r['Calls'])/3:6.1f}  avg_us {float(r['AverageNs'])/1e3:7.1f}  min {float(r['MinNs'])/1e3:7.1f}  max {float(r['MaxNs'])/1e3:7.1f}")
PY
  cd $R
  for k in 1 2 3; do python tools/bench_configs.py $([ $B = 256 ] && echo train27 || echo small) 2>/dev/null | grep '^{' | head -1 | cut -c1-120; done
done
unset KASF_LIB
