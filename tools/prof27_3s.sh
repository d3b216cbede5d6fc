#!/bin/bash
# three-stream kernel trace of three training steps (T=27, B=256) + the concurrency summary of the last one: bash tools/prof27_3s.sh [T] [B]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-27}; B=${2:-256}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof${T}_3s
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof${T}_3s -o t --output-format csv -- python3 $R/tools/train_once.py $T $B > $O/prof${T}_3s.log 2>&1; echo "prof rc=$?"
cd $R && python tools/overlap.py $O/prof${T}_3s/t_kernel_trace.csv
