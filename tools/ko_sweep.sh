#!/bin/bash
# hot-loop times of probe builds of the library: bash tools/ko_sweep.sh "<tag> <tag> ..."   (libkasf_hip_<tag>.so; "base" = the shipped build)
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/ko; mkdir -p $O
cd $R
for tag in $1; do
  if [ $tag = base ]; then unset KASF_LIB; else export KASF_LIB=$R/kasportsformer_amd/libkasf_hip_$tag.so; fi
  echo "== $tag: $(timeout -k 10 120 python tools/mlp_bench.py 2>&1 | tail -1)" | tee -a $O/sweep.txt
done
unset KASF_LIB
