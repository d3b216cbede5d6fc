#!/bin/bash
# One GPU-box pass that regenerates the round-3 evidence under gpurun_out/r3/ (copy into profiles/ afterwards: the names match).
# Order matters: the traces and counter passes come first and are copied into the box's profiles/ so that the bench line at the end quotes them.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; P=$R/profiles
mkdir -p $O; cd $R
say() { echo "== $(date +%T) $*"; }
say "three-stream traces"; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof27 -o t --output-format csv -- python3 $R/tools/train_once.py 27 256 > $O/prof27.log 2>&1 || echo "prof27 failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof81 -o t --output-format csv -- python3 $R/tools/train_once.py 81 128 > $O/prof81.log 2>&1 || echo "prof81 failed"
cp $O/prof27/t_kernel_stats.csv $O/r3_train_kernel_stats.csv; cp $O/prof81/t_kernel_stats.csv $O/r3_train81_kernel_stats.csv
say "evaluation-mode trace (forward only, B = 256)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/profe -o t --output-format csv -- python3 $R/tools/eval_once.py 256 > $O/profe.log 2>&1 || echo "profe failed"
cp $O/profe/t_kernel_stats.csv $O/r3_eval_kernel_stats.csv
say "single-stream traces (isolated launches)"
export KASF_SINGLE_STREAM=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof27s -o t --output-format csv -- python3 $R/tools/train_once.py 27 256 > $O/prof27s.log 2>&1 || echo "prof27s failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof81s -o t --output-format csv -- python3 $R/tools/train_once.py 81 128 > $O/prof81s.log 2>&1 || echo "prof81s failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof243s -o t --output-format csv -- python3 $R/tools/train_once.py 243 32 > $O/prof243s.log 2>&1 || echo "prof243s failed"
unset KASF_SINGLE_STREAM
cp $O/prof27s/t_kernel_stats.csv $O/r3_single_stream_kernel_stats.csv; cp $O/prof81s/t_kernel_stats.csv $O/r3_single_stream81_kernel_stats.csv; cp $O/prof243s/t_kernel_stats.csv $O/r3_single_stream243_kernel_stats.csv
cd $R
python tools/hbm_table.py $O/r3_single_stream_kernel_stats.csv 256 27 $O/r3_train_kernel_stats.csv > $O/r3_op_hbm.json
python tools/hbm_table.py $O/r3_single_stream81_kernel_stats.csv 128 81 $O/r3_train81_kernel_stats.csv > $O/r3_op_hbm_t81.json
say "whole-step HBM bytes"
bash tools/pmc_step.sh > $O/pmc_step.log 2>&1; cp gpurun_out/pmc_step.json $O/r3_pmc_step.json; tail -1 $O/pmc_step.log
say "MLP micro-benchmark: per-launch HBM bytes and durations"
cd /tmp
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 $R/tools/mlp_bench.py > $O/pmc_f.log 2>&1 || echo "fetch failed"
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 $R/tools/mlp_bench.py > $O/pmc_w.log 2>&1 || echo "write failed"
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/profm -o m --output-format csv -- python3 $R/tools/mlp_bench.py > $O/profm.log 2>&1 || echo "profm failed"
cd $R; python tools/pmc_traffic.py gpurun_out/r3/pmc_f gpurun_out/r3/pmc_w gpurun_out/r3/r3_pmc_traffic.json > /dev/null; cp $O/profm/m_kernel_stats.csv $O/r3_mlp_microbench_kernel_stats.csv
say "operator benches"
timeout -k 10 200 python tools/op_bench.py > $O/r3_op_bench.txt 2>&1
timeout -k 10 100 python tools/attn81_bench.py 2>/dev/null > $O/r3_attn81_bench.txt
say "other configurations"
: > $O/r3_configs.jsonl
for c in train27fp32 train81 train243 eval dropin; do timeout -k 10 400 python tools/bench_configs.py $c 2>/dev/null | grep '^{' >> $O/r3_configs.jsonl; done
KASF_SINGLE_STREAM=1 timeout -k 10 200 python tools/bench_configs.py train27 2>/dev/null | grep '^{' | sed 's/"config": "train/"config": "KASF_SINGLE_STREAM=1 train/' >> $O/r3_configs.jsonl
say "bench line (quotes the files above)"
cp $O/r3_train_kernel_stats.csv $O/r3_pmc_step.json $O/r3_pmc_traffic.json $P/
timeout -k 10 600 python bench.py > $O/r3_bench_b256.json 2> $O/bench.err; echo "bench rc=$?"
rm -rf $O/prof27 $O/prof81 $O/prof27s $O/prof81s $O/prof243s $O/profe $O/pmc_f $O/pmc_w $O/profm $R/gpurun_out/pmcs_f $R/gpurun_out/pmcs_w
say done; ls $O
