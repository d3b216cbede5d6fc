"""The symmetric ("lockstep") MLP kernels stay in the library as the comparison point for the specialised-wave kernels and are
selected by environment switches that the library reads once per process: run the MLP operator tests once more in a child process with the
switches set, so that both forms stay checked against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("switches", [{"KASF_MLP_FWD_LOCKSTEP": "1", "KASF_MLP_BWD_LOCKSTEP": "1"}, {"KASF_MLP_BWD_XCHG": "1"}])
def test_symmetric_mlp_kernels_still_match(switches):
    env = dict(os.environ, **switches)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_ops.py"), "-x", "-q", "-m", "gpu", "-k", "mlp"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


WORKER = r"""
import os, sys, torch
sys.path.insert(0, os.environ["KASF_ROOT"])
import kasportsformer_amd as K
torch.manual_seed(114514)
m = K.KASportsFormer(n_layers=1, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().train()
with torch.no_grad():                       # O(1) layer scales and biases: the terms under test are not multiplied by 1e-5
    for n, p in m.named_parameters():
        if "layer_scale" in n or n.endswith("fc2.bias"):
            p.copy_(torch.randn_like(p) * 0.5)
x, y = (t.cuda() for t in K.synthetic_clips(96, 27, seed=5))
out = {}
for rep in range(3):                        # three passes on the same scratch: the ticket word must come back to zero every time
    m.zero_grad(set_to_none=True); m.flat_grad = None
    K.loss3(m(x), y)[0].backward()
    torch.cuda.synchronize()
    for n, p in m.named_parameters():
        # attention and bone blocks only: their gradient stream has no atomics upstream (the graph blocks' BatchNorm sums do, and one flipped bf16
        # rounding there moves everything downstream by ~1e-3), so what is left is the fp32 noise of the sums under test
        if ("att_" in n or "bone_" in n) and (n.endswith("mlp.fc2.bias") or n.endswith("layer_scale_2") or n.endswith("mlp.fc2.weight") or n.endswith("mlp.fc1.weight")):
            out[f"{rep}:{n}"] = p.grad.detach().float().cpu().clone()
torch.save(out, sys.argv[1])
"""


def test_merged_mlp_finish_equals_the_two_launch_form(tmp_path):
    """k_lnbwd_sum4_fin (one launch: partial reductions beside the dA stream, colsum(g) terms applied by the last workgroup through a ticket) against
    k_lnbwd_sum4 + k_mlp_wfinish: the fc2 bias and layer-scale gradients -- the two tensors that depend on the complete colsum(g) -- agree to the
    noise of their fp32 atomics (a workgroup's share arriving after the ticket would be an error of ~1/512 of the sum), the weight gradients bit for bit."""
    import torch
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    files = []
    for tag, extra in (("merged", {}), ("separate", {"KASF_MLP_FINISH_SEPARATE": "1"})):
        f = tmp_path / f"{tag}.pt"
        env = dict(os.environ, KASF_ROOT=ROOT, **extra)
        out = subprocess.run([sys.executable, str(script), str(f)], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        files.append(torch.load(str(f)))
    a, b = files
    assert a.keys() == b.keys() and len(a) == 3 * 1 * 4 * 4
    worst = 0.0
    for k in a:
        scale = float(b[k].abs().max())
        assert scale > 0, k
        if k.endswith("weight"):
            assert torch.equal(a[k], b[k]), k           # fixed-order partial reductions: the same bits in both forms
        else:
            d = float((a[k] - b[k]).abs().max()) / scale
            worst = max(worst, d)
            assert d < 5e-5, (k, d)                      # (a lost share would be ~2e-3 in every channel)
    print("merged vs separate finish: worst relative difference of the colsum(g)-dependent gradients", worst)


def test_fused_attention_block_backward_still_matches():
    """k_attn_blk_bwd (KASF_ATTN_BLOCK_BWD=1: attention backward + QKV data gradient + LayerNorm backward in one role-specialised launch) is not the
    default but stays in the library: the model's stage / gradient tests once more with the switch set."""
    env = dict(os.environ, KASF_ATTN_BLOCK_BWD="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_model.py"), "-x", "-q", "-m", "gpu", "-k",
                          "stage_by_stage or training_step_with_fused_adamw or single_clip or (backward_matches_oracle and 2-27-2)"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout
