"""Which gradient tensors differ between two identical training steps?  (debugging aid for tests/test_gpu_determinism.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kasportsformer_amd as K
from oracle import kasf_oracle as O
from tests.gpu_util import make_pair

cd = sys.argv[1] if len(sys.argv) > 1 else "bf16"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1
T = int(sys.argv[3]) if len(sys.argv) > 3 else 27
_, model = make_pair(L, T, cd)
x, y = (t.cuda() for t in O.synthetic_clips(16, T, seed=91))
model.train()
buffers, nbt = model._flat_buffers.clone(), model._nbt.clone()
grads = []
for rep in range(2):
    model._flat_buffers.copy_(buffers); model._nbt.copy_(nbt)
    model.zero_grad()
    loss, _ = K.loss3(model(x), y)
    loss.backward()
    torch.cuda.synchronize()
    grads.append({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
bad = [(n, float((grads[0][n] - grads[1][n]).abs().max()), float(grads[0][n].abs().max())) for n in grads[0] if not torch.equal(grads[0][n], grads[1][n])]
print(f"{cd} L={L} T={T}: {len(bad)} of {len(grads[0])} gradient tensors differ")
for n, d, m in bad[:400]:
    print(f"  {n:70s} max diff {d:.3e} of {m:.3e}")
