"""Helpers shared by the GPU parity tests: build oracle + HIP model pairs, read workspace tensors."""
import ctypes as C

import numpy as np
import torch

from oracle import kasf_oracle as O

DT = {"fp32": (0, torch.float32), "bf16": (1, torch.bfloat16)}


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def rel_err(got: torch.Tensor, ref: torch.Tensor) -> float:
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))


def make_pair(n_layers, T, compute_dtype, salt=0, **kw):
    """Oracle (CPU fp32) and HIP model with identical, de-identitied, name-seeded parameters."""
    import kasportsformer_amd as K
    oracle = O.KASportsFormerOracle(n_layers=n_layers, num_heads=8, n_frames=T, **kw)
    sd = O.name_seeded_fill(oracle.state_dict(), salt)
    oracle.load_state_dict(sd, strict=True)
    model = K.KASportsFormer(n_layers=n_layers, num_heads=8, n_frames=T, compute_dtype=compute_dtype, **kw)
    model.load_state_dict(sd, strict=True)
    return oracle, model.cuda()


def ws_tensor(model, ws, batch, name, flags=1):
    """View of a named workspace activation (see kasf_ws_entry) as a torch tensor."""
    from kasportsformer_amd import _lib
    ents = _lib.ws_entries(model._device_handle(), batch, flags)
    off, numel, kind = ents[name]
    dt = {0: torch.bfloat16 if model.compute_dtype == "bf16" else torch.float32, 1: torch.float32, 2: torch.float64, 3: torch.int32}[kind]
    nbytes = numel * torch.empty((), dtype=dt).element_size()
    return ws[off:off + nbytes].view(dt)


def oracle_stage_hooks(oracle):
    """Registers forward hooks capturing every FormerModule / layer output; returns the dict they fill."""
    cap = {}
    for li, layer in enumerate(oracle.layers_with_bone):
        for kind in O.BLOCK_KINDS:
            getattr(layer, kind).register_forward_hook(lambda m, i, o, k=f"L{li}.{kind}.x_out": cap.__setitem__(k, o.detach()))
        layer.register_forward_hook(lambda m, i, o, k=f"L{li}.gate_out": cap.__setitem__(k, o.detach()))
    return cap


def decode_masks(words: torch.Tensor, G: int, T: int) -> torch.Tensor:
    """[G*T*MW] uint32 words (kasf_ws_entry 'adj_mask': row r of track g at ((g*T + r)*MW + w), bit c of word c>>5; MW = 3 up to 96 frames,
    ceil(T/32) beyond) -> bool [G,T,T]."""
    mw = 3 if T <= 96 else (T + 31) // 32
    w = words.cpu().view(G, T, mw).to(torch.int64) & 0xFFFFFFFF
    cols = torch.arange(T)
    return ((w[:, :, (cols >> 5)] >> (cols & 31)) & 1).bool()


class forced_adjacency:
    """Separates the path's DISCRETE decisions from its arithmetic.

    The temporal GCN keeps, per frame, every frame whose similarity reaches the 4th largest of the row (graph.py:104-112).  LN(x) rows of one
    joint track are nearly parallel, so the similarities of a row differ in the 3rd-4th digit and the 4th / 5th largest are often closer
    than bf16 -- sometimes closer than fp32 summation-order -- resolution.  One flipped neighbour is an O(1) change of that token, which the
    following layers carry on: comparing end-to-end outputs then measures luck, not kernels.  So: the adjacency bit masks the HIP forward
    stored are (1) compared row by row with the oracle's own `topk` decision -- every differing row must be a near-tie of the ORACLE's
    similarities in fp32 mode -- and (2) handed to the oracle in place of its own decision, after which outputs and gradients are compared
    at the arithmetic tolerance.  The adjacency carries no gradient (graph.py:81,111), so the backward comparison is unaffected."""

    def __init__(self, model, x, near_tie=1e-5):
        from kasportsformer_amd import _lib
        self.L, self.near_tie = model.n_layers, near_tie
        B, T = x.shape[0], x.shape[1]
        buf, nbt = model._flat_buffers.clone(), model._nbt.clone()
        with torch.no_grad():
            _, ws, _ = model._launch_forward(x.cuda(), False, keep=True)
        torch.cuda.synchronize()
        model._flat_buffers.copy_(buf); model._nbt.copy_(nbt)          # the probing forward must not leave a second running-statistics update behind
        self.masks = [decode_masks(ws_tensor(model, ws, B, f"L{li}.graph_temporal.adj_mask", flags=_lib.FLAG_TRAIN), B * 17, T) for li in range(self.L)]
        del ws
        self.calls = 0
        self.rows = self.mismatched = self.unexplained = self.bits = self.bits_equal = 0

    def _fn(self, g, k):
        forced = self.masks[self.calls % self.L]
        self.calls += 1
        sim = g.detach() @ g.detach().transpose(1, 2)
        top = sim.topk(min(k + 1, sim.shape[-1]), dim=-1)[0]
        natural = sim >= top[..., k - 1:k]
        if sim.shape[-1] > k:
            near = (top[..., k - 1] - top[..., k]).abs() <= self.near_tie * sim.abs().amax(dim=-1)
        else:                                                  # exactly k frames: every frame is a neighbour, nothing to tie with
            near = torch.zeros(sim.shape[:-1], dtype=torch.bool)
        bad = (natural != forced).any(dim=-1)
        self.rows += bad.numel(); self.mismatched += int(bad.sum()); self.unexplained += int((bad & ~near).sum())
        self.bits += natural.numel(); self.bits_equal += int((natural == forced).sum())
        return forced.to(g.dtype)

    def __enter__(self):
        self._orig = O.temporal_topk_adjacency
        O.temporal_topk_adjacency = self._fn
        return self

    def __exit__(self, *exc):
        O.temporal_topk_adjacency = self._orig
        return False

    def summary(self):
        return (f"adjacency rows {self.rows}: {100.0 * (self.rows - self.mismatched) / max(self.rows, 1):.3f} % identical to the oracle's own top-4 "
                f"({self.mismatched} differ, {self.unexplained} of them not near-ties); entries {100.0 * self.bits_equal / max(self.bits, 1):.4f} % equal")


def compare_grads(model, oracle, cd, floor_rel=None, small=256, pool_upto=64):
    """Every gradient of the HIP model against the oracle's.  Per-tensor error = max |g - r| / max(|r|max, floor_rel * gmax) (the floor: gradients that are
    sums of cancelling terms sit at the summation-noise floor of BOTH sides).  bf16 mode: tensors of at most `pool_upto` elements (the limb-MLP weights of
    16 ... 64 elements at the end of the longest path, biases of a few entries) carry a difference of large noisy terms -- their individual error is a sample
    of bf16 rounding noise, not a property of a kernel -- so they are judged POOLED (one cosine over all of them).  The tensors between that and `small`
    elements -- the 128-element LayerNorm gammas / betas, biases and layer scales, each the output of a per-workgroup-row reduction of some kernel -- are
    judged ONE BY ONE by their cosine (`mid_min_cosine`; ADVICE r4: a pooled cosine is dominated by the largest of them, a dropped or wrong small tensor
    could pass), and the max-error bar applies to the rest.
    Returns a dict: worst (error, name) over the per-tensor set, pooled cosine of the tiny set, worst per-tensor cosine of the middle set, cosine over everything."""
    if floor_rel is None:
        floor_rel = 1e-3 if cd == "fp32" else 0.05
    ref = dict(oracle.named_parameters())
    gmax = max(float(q.grad.abs().max()) for q in ref.values() if q.grad is not None)
    errs, pooled, dots, none_mismatch, mid = {}, [0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [], {}
    for n, p in model.named_parameters():
        r = ref[n].grad
        if (r is None) != (p.grad is None):
            none_mismatch.append(n)
            continue
        if r is None:
            continue
        g, r = p.grad.detach().double().cpu(), r.double()
        d = (float((g * r).sum()), float((g * g).sum()), float((r * r).sum()))
        for k in range(3):
            dots[k] += d[k]
        if cd == "bf16" and r.numel() < small:
            for k in range(3):
                pooled[k] += d[k]
            if r.numel() > pool_upto:
                mid[n] = d[0] / max(1e-300, d[1] ** 0.5 * d[2] ** 0.5) if d[2] > 0 else (1.0 if d[1] == 0 else 0.0)
            continue
        errs[n] = float((g - r).abs().max() / max(float(r.abs().max()), floor_rel * gmax))
    worst = max(errs, key=errs.get)
    mid_name = min(mid, key=mid.get) if mid else None
    return {"errors": errs, "worst": errs[worst], "worst_name": worst, "gmax": gmax, "none_mismatch": none_mismatch,
            "pooled_small_cosine": pooled[0] / max(1e-300, pooled[1] ** 0.5 * pooled[2] ** 0.5) if pooled[2] > 0 else 1.0,
            "mid_cosines": mid, "mid_min_cosine": mid[mid_name] if mid else 1.0, "mid_min_name": mid_name,
            "cosine": dots[0] / max(1e-300, dots[1] ** 0.5 * dots[2] ** 0.5)}
