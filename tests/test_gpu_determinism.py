"""same seed, same bits.  Two identical training steps from the same state must agree bit for bit -- predictions, loss terms,
BatchNorm running statistics and EVERY gradient -- in both arithmetic modes, and so must whole optimisation trajectories.

How: no gradient leaves a kernel through a floating-point atomic.  Per-channel sums (LayerNorm gamma / beta, biases, layer scales, the small
top-level tensors) are stored as one row per workgroup and added in a fixed order by one finishing launch per backward stage
(csrc/k_reduce.hip); GEMM weight gradients are per-split partial tiles + a fixed-order reduce; sums inside a workgroup use private LDS rows and
fixed trees instead of LDS atomics.  The only atomics left are the fp64 BatchNorm batch sums across workgroups, whose addends are fp32 numbers
of comparable size: every partial sum is exactly representable, so their order cannot matter either."""
import pytest
import torch

from oracle import kasf_oracle as O
from tests.gpu_util import make_pair

pytestmark = pytest.mark.gpu


def _step(K, model, x, y):
    model.flat_grad = None
    pred = model(x)
    loss, parts = K.loss3(pred, y)
    loss.backward()
    torch.cuda.synchronize()
    return pred.detach().clone(), parts.clone(), model.flat_grad[:model.n_live].clone(), model._flat_buffers.clone()


@pytest.mark.parametrize("T,B", [(27, 16), (81, 3)])
@pytest.mark.parametrize("cd", ["fp32", "bf16"])
def test_same_inputs_same_bits(cd, T, B):
    import kasportsformer_amd as K
    _, model = make_pair(3, T, cd)
    x, y = (t.cuda() for t in O.synthetic_clips(B, T, seed=91))
    model.eval()
    with torch.no_grad():
        outs = [model(x) for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    model.train()
    model.attach_param_grads = False
    buffers, nbt = model._flat_buffers.clone(), model._nbt.clone()
    runs = []
    for rep in range(3):
        model._flat_buffers.copy_(buffers); model._nbt.copy_(nbt)
        if rep == 2:                               # the third pass with other work in flight on another stream (different scheduling, same bits)
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                junk = torch.randn(4096, 4096, device="cuda")
                for _ in range(20):
                    junk = junk @ junk * 1e-3
        runs.append(_step(K, model, x, y))
    torch.cuda.synchronize()
    (p1, l1, g1, b1) = runs[0]
    assert float(g1.abs().max()) > 0
    for p2, l2, g2, b2 in runs[1:]:
        assert torch.equal(p1, p2), "predictions"
        assert torch.equal(l1, l2), "loss terms"
        assert torch.equal(b1, b2), "BatchNorm running statistics"
        same = float((g1 == g2).float().mean())
        assert torch.equal(g1, g2), f"gradients: {100 * same:.4f} % bit-identical, max diff {float((g1 - g2).abs().max()):.3e}"


def test_training_trajectory_is_reproducible():
    """Twenty optimisation steps of the 2-layer bf16 model twice from the same weights: identical parameters at the end, bit for bit."""
    import kasportsformer_amd as K
    _, model = make_pair(2, 27, "bf16")
    xs, ys = (t.cuda() for t in O.teacher_clips(8 * 4, 27, seed=93))
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    finals = []
    for _ in range(2):
        model.load_state_dict(sd0)
        model.train()
        model.attach_param_grads = False
        opt = K.FusedAdamW(model, lr=5e-4, weight_decay=0.01)
        for s in range(20):
            o = (s % 4) * 8
            opt.zero_grad()
            K.loss3(model(xs[o:o + 8]), ys[o:o + 8])[0].backward()
            opt.step()
        torch.cuda.synchronize()
        finals.append(model._flat.clone())
    assert torch.equal(finals[0], finals[1])
