#!/bin/bash
# kernel-trace statistics of a few training steps: T=27/B=256 (configs[1]) and T=81/B=128 (configs[3]); results under gpurun_out/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof27 -o t27 --output-format csv -- python3 $R/tools/train_once.py 27 256 > $O/prof27.log 2>&1; echo "prof27 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof81 -o t81 --output-format csv -- python3 $R/tools/train_once.py 81 128 > $O/prof81.log 2>&1; echo "prof81 rc=$?"
ls $O/prof27 $O/prof81
