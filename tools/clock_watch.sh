#!/bin/bash
# shader clock and package power while the B = 256 training step runs (is the step power-limited?): bash tools/clock_watch.sh [seconds=20]
R=${GRAFT_REPO_ROOT:?}; cd $R; N=${1:-20}
echo "== idle"; rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power" | head -6
python - <<'P' &
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, kasportsformer_amd as K
torch.manual_seed(114514)
model = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().train()
model.attach_param_grads = False
opt = K.FusedAdamW(model, lr=5e-4, weight_decay=0.01)
x, y = (t.cuda() for t in K.synthetic_clips(256, 27, seed=1234))
def step():
    opt.zero_grad(); loss, _ = K.loss3(model(x), y); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.time(); n = 0
while time.time() - t0 < float(os.environ.get("WATCH_S", "20")):
    for _ in range(20): step()
    torch.cuda.synchronize(); n += 20
    print(f"t={time.time()-t0:5.1f}s  {256*n/(time.time()-t0):7.0f} clips/s so far", flush=True)
P
PID=$!
sleep 12   # model build + warm-up
for k in $(seq 1 $((N / 2))); do
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ' '; echo
  sleep 2
done
wait $PID
