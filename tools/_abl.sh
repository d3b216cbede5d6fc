python tools/attn81_bench.py 2>&1 | grep "us"
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" 2>&1 | tail -2
python tools/bench_configs.py train81 2>&1 | tail -1
