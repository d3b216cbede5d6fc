#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4h; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_model.py -m gpu -q -s > $O/pytest_model.log 2>&1; echo "pytest model rc=$?"; tail -4 $O/pytest_model.log
python -m pytest tests/test_gpu_determinism.py tests/test_gpu_train_parity.py -m gpu -q -x > $O/pytest_det.log 2>&1; echo "pytest det rc=$?"; tail -2 $O/pytest_det.log
bash tools/prof27.sh 81 128 > $O/prof81.txt 2>&1; grep -i "agg_temporal\|sum of kernel" $O/prof81.txt
python tools/bench_configs.py train81 > $O/train81.jsonl 2>/dev/null; cat $O/train81.jsonl
SCRIPT=tools/mlp_bench.py bash tools/pmc_kernel.sh k_mlp_fwd_s 117504 > $O/pmc_mlp_fwd.txt 2>&1; cat $O/pmc_mlp_fwd.txt
SCRIPT=tools/mlp_bench.py bash tools/pmc_kernel.sh k_mlp_bwd_s 117504 > $O/pmc_mlp_bwd.txt 2>&1; cat $O/pmc_mlp_bwd.txt
