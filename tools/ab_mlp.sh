#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
run() { echo "== $1"; shift; env "$@" timeout -k 10 120 python tools/mlp_bench.py 2>/dev/null | tail -1 | cut -c1-110; }
run "default (sum4 chain, 16-byte partial stores)" X=1
run "xchg opt-in" KASF_MLP_BWD_XCHG=1
run "default (sum4 chain, 16-byte partial stores)" X=1
touch kasportsformer_amd/csrc/k_mlp3.hip && make -C kasportsformer_amd/csrc EXTRA="-DKASF_BWD_STORE8" 2>&1 | grep -E "error"
run "8-byte stores (round 1)" X=1
run "8-byte stores (round 1)" X=1
