#!/bin/bash
# same-box A/B of builds of the library: bash tools/ab.sh "<variant.so> [<variant2.so> ...]" [kernel regex]
# (variant builds: make -C kasportsformer_amd/csrc BUILD=build_x LIB=../libkasf_hip_x.so [EXTRA=-D...])
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; VS=$1; RX=${2:-.}; O=$R/gpurun_out/ab; mkdir -p $O
cd $R
for tag in base $VS; do
  if [ $tag = base ]; then unset KASF_LIB; else export KASF_LIB=$R/kasportsformer_amd/$tag; fi
  bash tools/prof27.sh 27 256 > $O/prof27_$tag.txt 2>&1; echo "== $tag: $(grep 'sum of kernel' $O/prof27_$tag.txt)"; grep -E "$RX" $O/prof27_$tag.txt | cut -c1-60,86-
  python tools/bench_configs.py train27 2>/dev/null | grep '^{'
done
unset KASF_LIB
