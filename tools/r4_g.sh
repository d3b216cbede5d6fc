#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4g; mkdir -p $O
cd $R
KASF_LIB=$R/kasportsformer_amd/libkasf_hip_shared.so python -m pytest tests/test_gpu_model.py -m gpu -q -x -s -k "backward_matches and bf16 and 2-27-2" > $O/pytest_shared.log 2>&1; echo "shared-streams variant (old aggregate) rc=$?"; grep "backward, bf16" $O/pytest_shared.log
python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "backward_matches and bf16" > $O/pytest_new.log 2>&1; echo "new rc=$?"; grep "backward, bf16" $O/pytest_new.log
python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "stage" > $O/pytest_stage.log 2>&1; echo "stage rc=$?"; grep "stages, \|passed\|failed" $O/pytest_stage.log | head
