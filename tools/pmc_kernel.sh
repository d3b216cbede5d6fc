#!/bin/bash
# SQ counter passes over a training run restricted to kernels matching a regex: bash tools/pmc_kernel.sh <regex> [T B]   -> prints per-launch means
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmck; RX=$1; T=${2:-81}; B=${3:-128}
rm -rf $O; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "$RX" -d $O/p$i -o p --output-format csv -- python3 $R/tools/train_once.py $T $B > $O/p$i.log 2>&1 || echo "set $i failed"
done
cd $R && python - <<PY
import collections, csv, glob
agg = collections.defaultdict(list); dur = []
for f in sorted(glob.glob("$O/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{k:28s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY
