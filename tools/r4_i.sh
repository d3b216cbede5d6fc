#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4i; mkdir -p $O
cd $R
export KASF_LIB=$R/kasportsformer_amd/libkasf_hip_agg2.so
python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "topk or stages or backward_matches or golden" > $O/pytest_agg2.log 2>&1; echo "pytest agg2 rc=$?"; tail -2 $O/pytest_agg2.log
bash tools/prof27.sh 81 128 > $O/prof81_agg2.txt 2>&1; grep -i "agg_temporal\|sum of kernel" $O/prof81_agg2.txt
bash tools/prof27.sh 27 256 > $O/prof27_agg2.txt 2>&1; grep -i "agg_temporal\|sum of kernel" $O/prof27_agg2.txt
