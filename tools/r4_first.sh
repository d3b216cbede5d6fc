#!/bin/bash
# round 4, first box pass: whole GPU suite, the per-rank regime of configs[2] (B = 32 / 64), a 3-stream + a 1-stream kernel trace of the B = 32 step, baseline bench
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4a; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python tools/bench_configs.py small > $O/small.jsonl 2> $O/small.err; echo "small rc=$?"; cat $O/small.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof32 -o t --output-format csv -- python3 $R/tools/train_once.py 27 32 > $O/prof32.log 2>&1; echo "prof32 rc=$?"
KASF_SINGLE_STREAM=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof32s -o t --output-format csv -- python3 $R/tools/train_once.py 27 32 > $O/prof32s.log 2>&1; echo "prof32s rc=$?"
cd $R
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
