#!/bin/bash
# Kernel-trace statistics of three training steps at T=27/B=256 with the three branches SERIALISED on one stream (KASF_SINGLE_STREAM=1): every
# kernel has the chip to itself, so the averages are isolated durations and their sum is the step a one-stream engine would run.
# -> gpurun_out/prof27s/ ; prints per-kernel isolated time per step, sorted.   usage: bash tools/prof27.sh [T] [B]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-27}; B=${2:-256}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof${T}s
export KASF_SINGLE_STREAM=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof${T}s -o t --output-format csv -- python3 $R/tools/train_once.py $T $B > $O/prof${T}s.log 2>&1; echo "prof rc=$?"
cd $R && python - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof${T}s/t_kernel_stats.csv")))
rows=[r for r in rows if 'rocclr_copyBuffer' not in r['Name']]     # the constructor's per-parameter copies into the flat array (setup, not the step)
tot=sum(float(r['TotalDurationNs']) for r in rows)/3e6
print(f"sum of kernel time per step (single stream): {tot:.2f} ms")
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:40]:
    print(f"{r['Name'][:84]:84s} n/step {int(r['Calls'])/3:6.1f}  ms/step {float(r['TotalDurationNs'])/3e6:6.2f}  avg_us {float(r['AverageNs'])/1e3:7.1f}  min_us {float(r['MinNs'])/1e3:7.1f}")
PY
