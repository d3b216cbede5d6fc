#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4j; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "topk or stages or backward_matches or golden or arbitrary or neighbour or zero_length" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python -m pytest tests/test_gpu_determinism.py -m gpu -q -x > $O/pytest_det.log 2>&1; echo "pytest det rc=$?"; tail -1 $O/pytest_det.log
bash tools/prof27.sh 81 128 > $O/prof81.txt 2>&1; grep -i "agg_temporal\|sum of kernel" $O/prof81.txt
bash tools/prof27.sh 27 256 > $O/prof27.txt 2>&1; grep -i "agg_temporal\|sum of kernel" $O/prof27.txt
python tools/bench_configs.py train81 > $O/train81.jsonl 2>/dev/null; cat $O/train81.jsonl
