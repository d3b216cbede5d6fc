#!/bin/bash
# same-box A/B of two builds of the library: bash tools/ab.sh <variant.so> [kernel regex]      (variant builds: make -C kasportsformer_amd/csrc BUILD=build_x LIB=../libkasf_hip_x.so [EXTRA=-D...])
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; V=$1; RX=${2:-.}; O=$R/gpurun_out/ab; mkdir -p $O
cd $R
for tag in base variant; do
  if [ $tag = variant ]; then export KASF_LIB=$R/kasportsformer_amd/$V; else unset KASF_LIB; fi
  bash tools/prof27.sh 27 256 > $O/prof27_$tag.txt 2>&1; echo "== $tag: $(grep 'sum of kernel' $O/prof27_$tag.txt)"; grep -E "$RX" $O/prof27_$tag.txt | cut -c1-60,86-
  python tools/bench_configs.py train27 2>/dev/null | grep '^{'
done
unset KASF_LIB
