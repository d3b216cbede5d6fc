#!/bin/bash
# VGPR / SGPR / scratch of every kernel of one source file: bash tools/regs.sh k_attn_blk.hip [grep pattern] [extra flags]
F=$1; P=${2:-.}; shift; shift
D=/root/repo/kasportsformer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -I/root/repo/include -I$D --cuda-device-only -S -o /tmp/regs_$$.s $D/$F "$@" 2>&1 | grep -E "error" 
grep -E "^\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|name):|\.vgpr_spill_count" /tmp/regs_$$.s | paste - - - - - | sed 's/ \+/ /g; s/\.private_segment_fixed_size/scratch/; s/\.vgpr_spill_count/spills/' | cut -c1-190 | grep -E "$P"
cp /tmp/regs_$$.s /tmp/regs_last.s; rm -f /tmp/regs_$$.s
