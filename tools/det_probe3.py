"""Three identical bf16 training steps with the three branch streams (3 layers, T = 27 and 81): are all gradients bit-identical?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import test_gpu_determinism as t
for T, B in ((27, 16), (81, 3)):
    for rep in range(4):
        runs = t._three_runs("bf16", T, B)
        g0 = runs[0][2]
        same = [float((g0 == r[2]).float().mean()) for r in runs[1:]]
        print(f"T={T} rep {rep}: fraction of gradient elements bit-identical to run 0: {same}", flush=True)
