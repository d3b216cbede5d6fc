"""Micro-benchmark of the MLP kernels only (for rocprofv3 --pmc runs)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 27 * 17
print(json.dumps({k: (round(v["seconds"] * 1e6, 1), round(v["achieved_tflops"], 1)) for k, v in bench.kernel_rooflines(M).items()}))
