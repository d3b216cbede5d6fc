#!/bin/bash
# SQ counter passes restricted to kernels matching a regex, per-launch means.
#   bash tools/pmc_kernel.sh <regex> [T B]            over tools/train_once.py T B
#   SCRIPT=tools/mlp_bench.py bash tools/pmc_kernel.sh <regex>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmck; RX=$1; T=${2:-81}; B=${3:-128}; S=${SCRIPT:-tools/train_once.py}
rm -rf $O; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "$RX" -d $O/p$i -o p --output-format csv -- python3 $R/$S $T $B > $O/p$i.log 2>&1 || echo "set $i failed: $(tail -2 $O/p$i.log | head -1)"
done
cd $R && python - <<PY
import collections, csv, glob
agg = collections.defaultdict(list)
for f in sorted(glob.glob("$O/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{k:32s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY
