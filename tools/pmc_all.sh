#!/bin/bash
# HBM traffic (two passes) and SQ counter passes over tools/mlp_bench.py; results under gpurun_out/
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_f -- python3 $R/tools/mlp_bench.py > $R/gpurun_out/pmc_f.log 2>&1 || echo "fetch failed"
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w -- python3 $R/tools/mlp_bench.py > $R/gpurun_out/pmc_w.log 2>&1 || echo "write failed"
cd $R && python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_traffic.json > /dev/null
bash tools/pmc_sq.sh
