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
import collections, csv, glob, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$O/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)(<[^(]*>)?", r["Kernel_Name"])          # one table per kernel (template arguments kept apart)
        agg[(m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
import json, os
if os.environ.get("OUT_JSON"):
    gm = lambda cs, n: (sum(cs[n]) / len(cs[n])) if n in cs else 0.0
    json.dump({kern: {"counters_mean_per_launch": {k: sum(v) / len(v) for k, v in cs.items()},
                      "mfma_busy": (gm(cs, "SQ_VALU_MFMA_BUSY_CYCLES") / (32 * gm(cs, "SQ_CYCLES"))) if gm(cs, "SQ_CYCLES") else None,
                      "lds_bank_conflict_frac": (gm(cs, "SQ_LDS_BANK_CONFLICT") / gm(cs, "SQ_LDS_IDX_ACTIVE")) if gm(cs, "SQ_LDS_IDX_ACTIVE") else None}
               for kern, cs in agg.items()}, open(os.environ["OUT_JSON"], "w"), indent=1)
for kern, cs in agg.items():
    print(f"== {kern}")
    for k, v in cs.items():
        print(f"{k:32s} {sum(v)/len(v):16.0f}   (n={len(v)})")
    g = lambda n: (sum(cs[n]) / len(cs[n])) if n in cs else 0.0
    if g("SQ_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
        # SQ_CYCLES is summed over the 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES over the 1,024 SIMDs: busy fraction of the matrix pipes over the launch
        print(f"   matrix pipe busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / (32 * g('SQ_CYCLES')):.3f} of the launch (all SIMDs); LDS bank conflicts {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.3f} of the LDS-active cycles")
    if g("SQ_WAVE_CYCLES") and g("SQ_WAVES"):
        wc = g("SQ_WAVE_CYCLES")
        print(f"   per wave: {4 * wc / g('SQ_WAVES'):.0f} cycles, {g('SQ_INSTS_VALU') / g('SQ_WAVES'):.0f} VALU, {g('SQ_INSTS_LDS') / g('SQ_WAVES'):.0f} LDS, {g('SQ_INSTS_MFMA') / g('SQ_WAVES'):.0f} MFMA instructions;"
              f"  of the wave-cycles: waiting {100 * g('SQ_WAIT_ANY') / wc:.0f} %, issue-stalled {100 * g('SQ_WAIT_INST_ANY') / wc:.0f} %, VALU active {100 * g('SQ_ACTIVE_INST_VALU') / wc:.0f} %")
PY
