#!/bin/bash
# One GPU-box pass that regenerates the evidence under profiles/ (copy the results from gpurun_out/ afterwards).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 300 python tools/bench_configs.py train81 2>/dev/null > $O/configs.jsonl; timeout -k 10 300 python tools/bench_configs.py eval 2>/dev/null >> $O/configs.jsonl
timeout -k 10 60 ./tools/valu_probe.bin 256 > $O/valu_probe.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o b --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline > $O/prof.log 2>&1; echo "prof rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/profm -o m --output-format csv -- python3 $R/tools/mlp_bench.py > $O/profm.log 2>&1; echo "profm rc=$?"
