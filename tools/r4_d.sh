#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4d; mkdir -p $O
cd $R
for s in F G; do timeout -k 10 300 python tools/dp_probe2.py $s >> $O/dp_probe3.jsonl 2>> $O/dp_probe3.err; echo "scenario $s rc=$?"; done
for s in A F G D; do KASF_LIB=$R/kasportsformer_amd/libkasf_hip_shared.so timeout -k 10 300 python tools/dp_probe2.py $s 2>> $O/dp_probe3.err | sed 's/"scenario": "/"scenario": "shared side streams, /' >> $O/dp_probe3.jsonl; echo "shared $s rc=$?"; done
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/dp_probe3.jsonl
# T = 81: the generic register-similarity aggregate against the T = 81 instantiation, isolated kernel times
bash tools/prof27.sh 81 128 > $O/prof81_default.txt 2>&1; grep -i "agg_temporal\|sum of kernel" $O/prof81_default.txt
KASF_LIB=$R/kasportsformer_amd/libkasf_hip_agg81g.so bash tools/prof27.sh 81 128 > $O/prof81_agg81g.txt 2>&1; grep -i "agg_temporal\|sum of kernel" $O/prof81_agg81g.txt
