#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4c; mkdir -p $O
cd $R
for s in A B C D E; do timeout -k 10 200 python tools/dp_probe2.py $s >> $O/dp_probe2.jsonl 2>> $O/dp_probe2.err; echo "scenario $s rc=$?"; done
GPU_MAX_HW_QUEUES=4 timeout -k 10 200 python tools/dp_probe2.py D 2>> $O/dp_probe2.err | sed 's/"scenario": "D"/"scenario": "D, GPU_MAX_HW_QUEUES=4"/' >> $O/dp_probe2.jsonl
GPU_MAX_HW_QUEUES=16 timeout -k 10 200 python tools/dp_probe2.py D 2>> $O/dp_probe2.err | sed 's/"scenario": "D"/"scenario": "D, GPU_MAX_HW_QUEUES=16"/' >> $O/dp_probe2.jsonl
cat $O/dp_probe2.jsonl
timeout -k 10 120 ./tools/valu_probe.bin 256 > $O/valu_probe.txt 2>&1; echo "valu rc=$?"; tail -8 $O/valu_probe.txt
timeout -k 10 300 python tools/parity_breakdown.py 7 2002 > $O/parity_7_2002.txt 2>&1; echo "parity rc=$?"; cat $O/parity_7_2002.txt
timeout -k 10 300 python tools/parity_breakdown.py 6 1001 > $O/parity_6_1001.txt 2>&1; head -12 $O/parity_6_1001.txt
