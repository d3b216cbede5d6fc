#!/bin/bash
# SQ counter passes over tools/mlp_bench.py (one rocprofv3 --pmc run per set); output under gpurun_out/pmcN
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_CYCLES"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc$i -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/mlp_bench.py > $GRAFT_REPO_ROOT/gpurun_out/pmc$i.log 2>&1 || echo "set $i failed"
done
ls $GRAFT_REPO_ROOT/gpurun_out
