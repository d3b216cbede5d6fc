"""Large batches on one GPU: evaluation at B = 2,048 (BASELINE configs[4] is 2,048 clips over 8 GPUs) against eight passes of 256 (must be the same bits), and
training at B = 512 (past every grid-width threshold).   python tools/big_batch_check.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import kasportsformer_amd as K
torch.manual_seed(114514)
model = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().eval()
x, _ = K.synthetic_clips(2048, 27, seed=1234)
x = x.cuda()
with torch.no_grad():
    y_big = model(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): model(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    parts = torch.cat([model(x[i:i + 256]) for i in range(0, 2048, 256)])
print("eval B=2048:", round(2048 / dt), "clips/s; finite", bool(torch.isfinite(y_big).all()), "; equals 8 x B=256 passes:", bool(torch.equal(y_big, parts)), float((y_big - parts).abs().max()))
model.train(); model.attach_param_grads = False
opt = K.FusedAdamW(model, lr=5e-4, weight_decay=0.01)
xb, yb = (t.cuda() for t in K.synthetic_clips(512, 27, seed=7))
for _ in range(2):
    opt.zero_grad(); loss, _ = K.loss3(model(xb), yb); loss.backward(); opt.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    opt.zero_grad(); loss, _ = K.loss3(model(xb), yb); loss.backward(); opt.step()
torch.cuda.synchronize()
print("train B=512:", round(512 * 3 / (time.perf_counter() - t0)), "clips/s; loss", float(loss), "finite grads", bool(torch.isfinite(model.flat_grad).all()))
