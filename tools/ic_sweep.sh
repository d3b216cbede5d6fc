#!/bin/bash
# Is the dA-partial round trip of the MLP backward (k_mlp_bwd_s writes 4 bf16 partials, k_lnbwd_sum4_fin reads them back: 2 x 1,024 B per token) HBM traffic or
# Infinity-Cache traffic?  The memory-side counters (FETCH_SIZE / WRITE_SIZE) count Infinity-Cache hits as well (MI355X_MICROARCH.md), so: a SIZE SWEEP.  The
# partials of one launch are 1,024 B x M; the 256 MiB cache holds them up to M ~ 260 k tokens if nothing else competed (x, g, LN(x), the outputs: 1,280 B per token
# more).  Per-token time of the two kernels against M says where residency ends.   bash tools/ic_sweep.sh > gpurun_out/r5_ic_sweep.txt
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/ic; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for M in 29376 58752 117504 176256 235008 352512 470016 705024; do
  rm -rf $O/m$M
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/m$M -o t --output-format csv -- python3 $R/tools/mlp_bench.py $M > $O/m$M.log 2>&1 || { echo "M=$M failed"; continue; }
  python3 - $O/m$M/t_kernel_stats.csv $M <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); M = int(sys.argv[2])
pick = lambda k: next((float(r["AverageNs"]) for r in rows if k in r["Name"]), float("nan"))
b, l, f = pick("k_mlp_bwd_s"), pick("k_lnbwd_sum4_fin"), pick("k_mlp_fwd_s")
print(f"M {M:7d}  partials {M * 1024 / 2**20:6.0f} MiB   k_mlp_bwd_s {b / 1e3:7.1f} us = {b / M:6.3f} ns/token   k_lnbwd_sum4_fin {l / 1e3:7.1f} us = {l / M:6.3f} ns/token "
      f"({M * 1792 / l:6.2f} GB/s x1e0 of its 1,792 B/token)   k_mlp_fwd_s {f / 1e3:7.1f} us = {f / M:6.3f} ns/token")
PY
done
