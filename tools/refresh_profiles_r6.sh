#!/bin/bash
# Before the gpurun call: `git rev-parse HEAD > .git_rev` (the box has no .git; the stamps quote that file as information -- the sha256 of the sources is what is checked).
# ONE GPU-box pass that regenerates the round-6 evidence under gpurun_out/r6/ (copy into profiles/ afterwards: the names match).
# Order matters: the traces and counter passes come first and are copied into the box's profiles/ so that the bench line at the end quotes THIS build's
# numbers.  Any failed trace aborts the pass (the round-3 script only echoed and went on, so a stale file could have ended up in a "final build" set).
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
[ -f "$R/bench.py" ] || { echo "refresh_profiles_r6: $R is not the repository root"; exit 2; }
export GRAFT_REPO_ROOT=$R          # the helper scripts (tools/pmc_step.sh ...) read it
O=$R/gpurun_out/r6; P=$R/profiles
mkdir -p "$O"; cd "$R"
PART=${1:-all}        # "1": traces and whole-step bytes; "2": MLP counters, operator benches, other configurations, bench lines; "all": both (needs ~25 min: over one gpurun call)
stamp_all() {         # every r6_* file says which sources it was measured on (tools/stamp.py): json / jsonl carry the stamp, csv / txt go into r6_STAMP.json
    for f in $O/r6_*.json $O/r6_*.jsonl; do [ -s "$f" ] && python tools/stamp.py --embed "$f"; done
    python tools/stamp.py --sidecar "$O" r6
}
say() { echo "== $(date +%T) $*"; }
need() { [ -s "$1" ] || { echo "refresh_profiles_r6: missing or empty $1 -- aborting, nothing is copied into profiles/"; exit 3; }; }
trace() {   # trace <dir> <script args...>: kernel trace + stats of one python tool
    local d=$1; shift
    rm -rf "$O/$d"
    (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$O/$d" -o t --output-format csv -- python3 "$@" > "$O/$d.log" 2>&1)
    need "$O/$d/t_kernel_stats.csv"
}
if [ "$PART" = 1 ] || [ "$PART" = all ]; then
say "three-stream traces"
trace prof27 $R/tools/train_once.py 27 256;  cp $O/prof27/t_kernel_stats.csv $O/r6_train_kernel_stats.csv
trace prof81 $R/tools/train_once.py 81 128;  cp $O/prof81/t_kernel_stats.csv $O/r6_train81_kernel_stats.csv
trace prof32 $R/tools/train_once.py 27 32;   cp $O/prof32/t_kernel_stats.csv $O/r6_train_b32_kernel_stats.csv
say "evaluation-mode trace (forward only, B = 256)"
trace profe $R/tools/eval_once.py 256;       cp $O/profe/t_kernel_stats.csv $O/r6_eval_kernel_stats.csv
say "single-stream traces (isolated launches)"
export KASF_SINGLE_STREAM=1
trace prof27s $R/tools/train_once.py 27 256; cp $O/prof27s/t_kernel_stats.csv $O/r6_single_stream_kernel_stats.csv
trace prof81s $R/tools/train_once.py 81 128; cp $O/prof81s/t_kernel_stats.csv $O/r6_single_stream81_kernel_stats.csv
trace prof32s $R/tools/train_once.py 27 32;  cp $O/prof32s/t_kernel_stats.csv $O/r6_single_stream_b32_kernel_stats.csv
# the same one-stream steps with every persistent launch at its FULL grid (the engine's default gives the MLP launches half the chip below 150,000 tokens:
# kernels.h, kasf_narrow_grid): the figure that compares with rounds 1-3, where a launch alone on the chip had all 256 CUs
KASF_NARROW_PCTS=100,100,100,100,100,100,100 trace prof27sf $R/tools/train_once.py 27 256; cp $O/prof27sf/t_kernel_stats.csv $O/r6_single_stream_fullwidth_kernel_stats.csv
unset KASF_SINGLE_STREAM
say "concurrency of the three-stream step (tools/overlap.py on the trace above) and the same step with full-width launches"
python tools/overlap.py $O/prof27/t_kernel_trace.csv > $O/r6_overlap.txt; need $O/r6_overlap.txt
KASF_NARROW_PCTS=100,100,100,100,100,100,100 trace prof27f $R/tools/train_once.py 27 256
{ echo; echo "## KASF_NARROW_PCTS=100,100,100,100,100,100,100 (every launch at its full grid)"; python tools/overlap.py $O/prof27f/t_kernel_trace.csv; } >> $O/r6_overlap.txt
python tools/hbm_table.py $O/r6_single_stream_kernel_stats.csv 256 27 $O/r6_train_kernel_stats.csv > $O/r6_op_hbm.json;       need $O/r6_op_hbm.json
python tools/hbm_table.py $O/r6_single_stream81_kernel_stats.csv 128 81 $O/r6_train81_kernel_stats.csv > $O/r6_op_hbm_t81.json; need $O/r6_op_hbm_t81.json
say "whole-step HBM bytes"
bash tools/pmc_step.sh > $O/pmc_step.log 2>&1; need gpurun_out/pmc_step.json; cp gpurun_out/pmc_step.json $O/r6_pmc_step.json; tail -1 $O/pmc_step.log
stamp_all
fi
if [ "$PART" = 2 ] || [ "$PART" = all ]; then
# (run separately, part 2 finds part 1's files under profiles/ -- copied there and stamped by the caller after part 1 -- and leaves them alone)
for f in r6_train_kernel_stats.csv r6_single_stream_kernel_stats.csv r6_single_stream_fullwidth_kernel_stats.csv r6_pmc_step.json; do [ -s $O/$f ] || { [ -s $P/$f ] && cp $P/$f $O/; } || true; done
say "MLP micro-benchmark: per-launch HBM bytes and durations"
rm -rf $O/pmc_f $O/pmc_w $O/profm
(cd /tmp && TMPDIR=/tmp timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 $R/tools/mlp_bench.py > $O/pmc_f.log 2>&1)
(cd /tmp && TMPDIR=/tmp timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 $R/tools/mlp_bench.py > $O/pmc_w.log 2>&1)
(cd /tmp && TMPDIR=/tmp timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/profm -o m --output-format csv -- python3 $R/tools/mlp_bench.py > $O/profm.log 2>&1)
need $O/profm/m_kernel_stats.csv
python tools/pmc_traffic.py gpurun_out/r6/pmc_f gpurun_out/r6/pmc_w gpurun_out/r6/r6_pmc_traffic.json > /dev/null; need $O/r6_pmc_traffic.json
cp $O/profm/m_kernel_stats.csv $O/r6_mlp_microbench_kernel_stats.csv
say "SQ counters of the MLP kernels (matrix-pipe busy fraction, LDS bank conflicts): six --pmc passes over the micro-benchmark"
SCRIPT=tools/mlp_bench.py OUT_JSON=$O/r6_mlp_sq_counters.json bash tools/pmc_kernel.sh "k_mlp_fwd_s|k_mlp_bwd_s|k_lnbwd_sum4" 117504 > $O/r6_mlp_sq_counters.txt 2>&1; need $O/r6_mlp_sq_counters.json
say "operator benches"
timeout -k 10 200 python tools/op_bench.py > $O/r6_op_bench.txt 2>&1; need $O/r6_op_bench.txt
say "other configurations"
: > $O/r6_configs.jsonl
for c in train27fp32 train81 train243 eval dropin small; do timeout -k 10 400 python tools/bench_configs.py $c 2>/dev/null | grep '^{' >> $O/r6_configs.jsonl; done
KASF_NARROW_PCTS=100,100,100,100,100,100,100 timeout -k 10 200 python tools/bench_configs.py train27 2>/dev/null | grep '^{' | sed 's/"config": "train/"config": "KASF_NARROW_PCTS=100,... (full-width launches) train/' >> $O/r6_configs.jsonl
timeout -k 10 200 python tools/bench_configs.py train27 2>/dev/null | grep '^{' >> $O/r6_configs.jsonl
KASF_SINGLE_STREAM=1 timeout -k 10 200 python tools/bench_configs.py train27 2>/dev/null | grep '^{' | sed 's/"config": "train/"config": "KASF_SINGLE_STREAM=1 train/' >> $O/r6_configs.jsonl
need $O/r6_configs.jsonl
say "strong-scaling bench lines at the per-rank shapes of BASELINE configs[2] (one global batch of 256 over 8 / 4 ranks), data-parallel path, one rank"
timeout -k 10 300 python bench.py --force-dp --global-batch 32 --det-conf --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-kernel-roofline > $O/r6_bench_strong_b32.json 2> $O/bench32.err; need $O/r6_bench_strong_b32.json
timeout -k 10 300 python bench.py --force-dp --global-batch 64 --det-conf --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-kernel-roofline > $O/r6_bench_strong_b64.json 2> $O/bench64.err; need $O/r6_bench_strong_b64.json
say "26-layer parity samples (tests/test_gpu_model.py writes gpurun_out/r6_parity_26layers_{fp32,bf16}.json; the bench line's parity object is read from them)"
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "full_depth_26" > $O/parity26.log 2>&1 || { tail -5 $O/parity26.log; echo "refresh_profiles_r6: the 26-layer parity test failed"; exit 4; }
need gpurun_out/r6_parity_26layers_bf16.json
say "bench line (quotes the files above)"
for f in r6_parity_26layers_fp32.json r6_parity_26layers_bf16.json; do [ -s gpurun_out/$f ] && cp gpurun_out/$f $O/ || true; done
stamp_all
for f in r6_train_kernel_stats.csv r6_single_stream_kernel_stats.csv r6_single_stream_fullwidth_kernel_stats.csv r6_pmc_step.json r6_pmc_traffic.json r6_mlp_sq_counters.json r6_parity_26layers_fp32.json r6_parity_26layers_bf16.json; do
  [ -s $O/$f ] && cp $O/$f $P/ || true; done
python - <<'PY'
# the box's profiles/r6_STAMP.json must list the csv files just copied (the bench line checks them through it)
import json, os, shutil
shutil.copy("gpurun_out/r6/r6_STAMP.json", "profiles/r6_STAMP.json")
PY
timeout -k 10 600 python bench.py > $O/r6_bench_b256.json 2> $O/bench.err; need $O/r6_bench_b256.json
rm -rf $O/prof27 $O/prof27f $O/prof27sf $O/prof81 $O/prof32 $O/prof27s $O/prof81s $O/prof32s $O/profe $O/pmc_f $O/pmc_w $O/profm $R/gpurun_out/pmcs_f $R/gpurun_out/pmcs_w
stamp_all
say done; ls $O

fi
