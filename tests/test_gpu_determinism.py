"""same seed, same bits?  What is reproducible bit for bit and what is not (and by how much), measured on one training step.

Bitwise reproducible by construction: every evaluation-mode forward (no reductions across workgroups), the optimizer step, the clip
gather / flip, the evaluation metrics, and -- given a bit-identical gradient stream -- every GEMM weight gradient (per-split partial tiles
summed in a fixed order, never atomics).  NOT bitwise reproducible: reductions that end in floating-point atomics -- BatchNorm batch sums
(fp32 LDS atomics inside a workgroup, fp64 atomics across workgroups), LayerNorm / bias / layer-scale gradients (one fp32 atomic per channel
per workgroup).  The BatchNorm sums feed the training-mode forward, so two identical training steps agree to fp32 summation-order noise, not
to the bit.  This test states the sizes so that a change that makes things worse shows up."""
import pytest
import torch

from oracle import kasf_oracle as O
from tests.gpu_util import make_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cd", ["fp32", "bf16"])
def test_same_inputs_same_bits(cd):
    import kasportsformer_amd as K
    _, model = make_pair(3, 27, cd)
    x, y = (t.cuda() for t in O.synthetic_clips(16, 27, seed=91))
    # evaluation mode: bit for bit, every time
    model.eval()
    with torch.no_grad():
        outs = [model(x) for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # one training step twice from the same state
    model.train()
    model.attach_param_grads = False
    buffers, nbt = model._flat_buffers.clone(), model._nbt.clone()
    runs = []
    for _ in range(2):
        model._flat_buffers.copy_(buffers); model._nbt.copy_(nbt)
        model.flat_grad = None
        pred = model(x)
        loss, parts = K.loss3(pred, y)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((pred.detach().clone(), parts.clone(), model.flat_grad[:model.n_live].clone(), model._flat_buffers.clone()))
    (p1, l1, g1, b1), (p2, l2, g2, b2) = runs
    gmax = float(g1.abs().max())
    same_pred = float((p1 == p2).float().mean())
    same_grad = float((g1 == g2).float().mean())
    dpred = float((p1 - p2).abs().max() / p1.abs().max())
    dgrad = float((g1 - g2).abs().max() / gmax)
    dbuf = float((b1 - b2).abs().max() / b1.abs().max())
    print(f"[{cd}] two identical training steps: predictions {100 * same_pred:.3f} % bit-identical (max rel diff {dpred:.2e}); "
          f"gradients {100 * same_grad:.3f} % bit-identical (max diff {dgrad:.2e} of the largest gradient); BatchNorm running stats rel diff {dbuf:.2e}; "
          f"loss terms {[float(v) for v in (l1 - l2).abs()]}")
    tol = 1e-5 if cd == "fp32" else 2e-2          # bf16: an fp32-noise-sized change of a BatchNorm coefficient can move a bf16 activation by one ulp (2^-8)
    assert dpred <= tol and dgrad <= 10 * tol and dbuf <= 1e-5
    assert float((l1 - l2).abs().max()) <= tol * max(1.0, float(l1.abs().max()))
    # the optimizer step itself is elementwise: identical inputs, identical bits
    flat0 = model._flat.clone()
    res = []
    for _ in range(2):
        model._flat.copy_(flat0)
        model.flat_grad = torch.zeros(model.n_flat, device="cuda")
        model.flat_grad[:model.n_live].copy_(g1)
        opt = K.FusedAdamW(model, lr=5e-4, weight_decay=0.01)
        opt.step()
        res.append(model._flat.clone())
    assert torch.equal(res[0], res[1])
