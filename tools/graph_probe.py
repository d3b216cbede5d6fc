"""Does the eval-mode forward survive hipGraph capture (three engine streams joined by events), is the replay bit-identical, and what does it
buy at small batch?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, kasportsformer_amd as K
torch.manual_seed(0)
m = K.KASportsFormer(n_layers=26, num_heads=8, n_frames=27, compute_dtype="bf16").cuda().eval()
for B in (8, 32, 64, 256):
    x = K.synthetic_clips(B, 27, seed=3)[0].cuda()
    xs = x.clone()
    with torch.no_grad():
        ref = m(x).clone()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): m(xs)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(xs)
        def eager():
            return m(x)
        def graphed():
            xs.copy_(x); g.replay(); return out
        graphed(); torch.cuda.synchronize()
        same = bool((out == ref).all())
        res = []
        for fn in (eager, graphed):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): fn()
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"B={B}: bit-identical {same}; eager {res[0]:.2f} ms ({B / res[0] * 1e3:.0f} clips/s), graph {res[1]:.2f} ms ({B / res[1] * 1e3:.0f} clips/s)", flush=True)
