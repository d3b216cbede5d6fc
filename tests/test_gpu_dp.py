"""The data-parallel path on a real GPU with RCCL (`nccl` backend), world size 1: stage-sliced backward, one asynchronous all-reduce per
backward stage, finish_gradients, FusedAdamW with grad_scale.  With a single rank the all-reduce is the identity, so the run must
reproduce the plain (non-distributed) run up to the run-to-run noise of the fp32 atomics in the per-channel reductions -- which checks that
slicing the backward into stages and handing gradient buckets to RCCL on its own stream changes nothing.  (The N > 1 arithmetic is covered
on CPU by tests/test_dp_gloo.py.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["KASF_ROOT"])
import kasportsformer_amd as K
from oracle import kasf_oracle as O
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
def run(use_dp):
    torch.manual_seed(7)
    m = K.KASportsFormer(n_layers=3, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().train()
    m.attach_param_grads = False
    opt = K.FusedAdamW(m, lr=5e-4, weight_decay=0.01)
    dp = K.DataParallel(m, optimizer=opt) if use_dp else None
    assert dp is None or opt.grad_scale == 1.0 / dist.get_world_size()
    x, y = (t.cuda() for t in O.synthetic_clips(4, 27, seed=3))
    grads = None
    for step in range(3):
        opt.zero_grad()
        loss, _ = K.loss3(m(x), y)
        loss.backward()
        if dp is not None:
            dp.finish_gradients()
        if step == 0:
            grads = m.flat_grad[:m.n_live].clone()
        opt.step()
    torch.cuda.synchronize()
    return grads, m._flat.clone(), m._flat_buffers.clone(), float(loss.detach())
a, b, c = run(True), run(False), run(False)
gmax = float(b[0].abs().max())
noise = float((b[0] - c[0]).abs().max()) / gmax                      # two plain runs: atomics order
d = float((a[0] - b[0]).abs().max()) / gmax
assert d <= max(4 * noise, 1e-5), (d, noise)
assert float((a[1] - b[1]).abs().max()) <= max(4 * float((b[1] - c[1]).abs().max()), 1e-3), "parameters after 3 steps"
assert abs(a[3] - b[3]) <= 1e-3 * abs(b[3])
print("DP_OK", a[3], d, noise)
dist.destroy_process_group()
'''


def test_rccl_single_rank_matches_plain_run(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "dp_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, KASF_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="8")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "DP_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
