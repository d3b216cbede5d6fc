#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r4e; mkdir -p $O
cd $R
for s in G D E; do timeout -k 10 300 python tools/dp_probe2.py $s 2>> $O/dp_probe4.err | sed 's/"scenario": "/"scenario": "process-wide side streams (default build), /' >> $O/dp_probe4.jsonl; echo "$s rc=$?"; done
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/dp_probe4.jsonl
python tools/bench_configs.py small > $O/small.jsonl 2> $O/small.err; echo "small rc=$?"; grep config $O/small.jsonl
timeout -k 10 120 ./tools/packed_fp32_repro_slp.bin > $O/packed_repro_slp.txt 2>&1; echo "repro slp rc=$?"; cat $O/packed_repro_slp.txt
timeout -k 10 120 ./tools/packed_fp32_repro_noslp.bin > $O/packed_repro_noslp.txt 2>&1; echo "repro noslp rc=$?"; cat $O/packed_repro_noslp.txt
bash tools/pmc_kernel.sh k_gcn_agg_temporal 81 128 > $O/pmc_agg81.txt 2>&1; cat $O/pmc_agg81.txt
