#!/bin/bash
# Same-box A/B of the MLP kernel variants (tools/mlp_bench.py: isolated launches at M = 117,504): environment switches first, then compile-time ones.
R=$GRAFT_REPO_ROOT; cd $R
run() { echo "== $1"; shift; env "$@" timeout -k 10 120 python tools/mlp_bench.py 2>/dev/null | tail -1 | cut -c1-130; }
run "default (k_mlp_bwd_s + k_lnbwd_sum4_fin)" X=1
run "finish as a separate launch" KASF_MLP_FINISH_SEPARATE=1
run "in-kernel partial reduction (experimental)" KASF_MLP_BWD_XCHG=1
run "symmetric (lockstep) kernels" KASF_MLP_FWD_LOCKSTEP=1 KASF_MLP_BWD_LOCKSTEP=1
for v in "-DKASF_FWD_GELU_POLY=0" "-DKASF_BWD_STORE16"; do
  touch kasportsformer_amd/csrc/k_mlp3.hip && make -C kasportsformer_amd/csrc EXTRA="$v" 2>&1 | grep -E "error"
  run "$v" X=1
done
touch kasportsformer_amd/csrc/k_mlp3.hip && make -C kasportsformer_amd/csrc 2>&1 | grep -E "error"
