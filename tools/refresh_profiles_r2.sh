#!/bin/bash
# One GPU-box pass that regenerates the round-2 evidence under gpurun_out/ (copy into profiles/ afterwards).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_model.py -q -m gpu -k "head_counts or bare_constructor or arbitrary" > $O/r2_heads.log 2>&1; echo "heads/T tests rc=$?"; tail -2 $O/r2_heads.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r2_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r2_smoke.log
timeout -k 10 500 python bench.py > $O/r2_bench.json 2> $O/r2_bench.err; echo "bench rc=$?"
timeout -k 10 200 python tools/op_bench.py > $O/r2_op_bench.txt 2>&1; echo "op_bench rc=$?"
bash tools/prof_steps.sh
python tools/hbm_table.py $O/prof27/t27_kernel_stats.csv 256 27 > $O/r2_op_hbm.json
python tools/hbm_table.py $O/prof81/t81_kernel_stats.csv 128 81 > $O/r2_op_hbm_t81.json
cd /tmp && export TMPDIR=/tmp
for v in "" "KASF_MLP_BWD_XCHG=1"; do
  tag=${v:+_xchg}
  env $v timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f$tag -- python3 $R/tools/mlp_bench.py > $O/pmc_f$tag.log 2>&1 || echo "fetch failed"
  env $v timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w$tag -- python3 $R/tools/mlp_bench.py > $O/pmc_w$tag.log 2>&1 || echo "write failed"
  (cd $R && python tools/pmc_traffic.py gpurun_out/pmc_f$tag gpurun_out/pmc_w$tag gpurun_out/r2_pmc_traffic$tag.json > /dev/null)
done
env KASF_MLP_BWD_XCHG=1 timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/profm_x -o m --output-format csv -- python3 $R/tools/mlp_bench.py > $O/profm_x.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/profm -o m --output-format csv -- python3 $R/tools/mlp_bench.py > $O/profm.log 2>&1
echo done
