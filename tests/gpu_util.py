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


def make_pair(n_layers, T, compute_dtype, salt=0):
    """Oracle (CPU fp32) and HIP model with identical, de-identitied, name-seeded parameters."""
    import kasportsformer_amd as K
    oracle = O.KASportsFormerOracle(n_layers=n_layers, num_heads=8, n_frames=T)
    sd = O.name_seeded_fill(oracle.state_dict(), salt)
    oracle.load_state_dict(sd, strict=True)
    model = K.KASportsFormer(n_layers=n_layers, num_heads=8, n_frames=T, compute_dtype=compute_dtype)
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
