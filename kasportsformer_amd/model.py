"""Drop-in ``KASportsFormer`` for MI355X: same constructor, ``forward(x, return_rep=False)`` and
``state_dict`` layout as the reference module (model/KASportsFormer.py:290-347), with everything from
``forward`` downward executed by libkasf_hip (hand-written gfx950 kernels behind the C-ABI in
include/kasf.h).  PyTorch only owns device memory, streams and autograd plumbing here.

There is no CPU / eager fallback: calling ``forward`` on a CPU tensor, or importing without the built
shared library, raises.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import torch
from torch import nn

from . import _lib

BLOCK_KINDS = ("att_spatial", "att_temporal", "graph_spatial", "graph_temporal", "bone_spatial", "bone_temporal")
# modules/bone_refusion.py:34-40 (sizes only; the indices live in the kernels)
_LIMB_SIZES = (3, 3, 2, 2, 3, 3, 4, 4, 4, 4, 3, 4, 4, 4, 4, 2, 2)


class _Box(nn.Module):
    """Parameter container: mirrors a reference sub-module's registration order, has no forward."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("kasportsformer_amd sub-modules hold parameters only; call the top-level KASportsFormer (intermediate activations: model.stage_outputs(x))")


def _mlp_box(d_in, d_hidden, d_out):
    b = _Box()
    b.fc1 = nn.Linear(d_in, d_hidden)
    b.fc2 = nn.Linear(d_hidden, d_out)
    return b


def _mixer_box(kind, dim, nodes):
    b = _Box()
    if kind == "att":          # modules/selfattention.py:13-15 (proj is registered before qkv)
        b.proj = nn.Linear(dim, dim)
        b.qkv = nn.Linear(dim, dim * 3, bias=False)
    elif kind == "bone":       # modules/bone_crossattention.py:11-15
        b.proj = nn.Linear(dim, dim)
        b.qkv_q = nn.Linear(dim, dim, bias=False)
        b.qkv_kv = nn.Linear(dim, dim * 2, bias=False)
    else:                      # modules/graph.py:35-37,46-50
        b.U = nn.Linear(dim, dim)
        b.V = nn.Linear(dim, dim)
        b.batch_norm = nn.BatchNorm1d(nodes)
        std = (2.0 / dim) ** 0.5
        b.U.weight.data.normal_(0, std)
        b.V.weight.data.normal_(0, std)
    return b


def _former_box(kind, dim, n_frames, mlp_ratio, ls_init, use_layer_scale=True):
    mixer, mode = kind.split("_")           # model/KASportsFormer.py:65-101
    b = _Box()
    b.norm1 = nn.LayerNorm(dim)
    b.norm1_limb = nn.LayerNorm(dim)
    b.mixer = _mixer_box(mixer, dim, 17 if mode == "spatial" else n_frames)
    b.norm2 = nn.LayerNorm(dim)
    b.mlp = _mlp_box(dim, int(dim * mlp_ratio), dim)
    if use_layer_scale:                     # KASportsFormer.py:98-101; without it the block is x + mixer(..), x + mlp(..): the kernels then see the constant 1
        b.layer_scale_1 = nn.Parameter(ls_init * torch.ones(dim))
        b.layer_scale_2 = nn.Parameter(ls_init * torch.ones(dim))
    return b


class _KasfFunction(torch.autograd.Function):
    """Whole-model autograd node: forward and backward are single C-ABI calls."""

    @staticmethod
    def forward(ctx, x, anchor, model, return_rep):
        out, ws, flags = model._launch_forward(x, return_rep=return_rep, keep=True)
        ctx.model, ctx.ws, ctx.batch, ctx.flags = model, ws, x.shape[0], flags
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.ws is None:
            raise RuntimeError("Trying to backward through the graph a second time (the kept activations were freed)")
        ctx.model._launch_backward(ctx.ws, dout, ctx.batch, ctx.flags)
        ctx.ws = None
        return None, None, None, None


class KASportsFormer(nn.Module):
    """MI355X-native KASportsFormer.  Signature follows model/KASportsFormer.py:291-295; the extra
    keyword ``compute_dtype`` selects ``"bf16"`` (fast: bf16 MFMA, fp32 accumulate / statistics / master
    weights) or ``"fp32"`` (parity: exact-f32 MFMA)."""

    def __init__(self, n_layers=26, dim_in=3, dim_feat=128, dim_rep=512, dim_out=3, mlp_ratio=4, act_layer=nn.GELU, attn_drop=0.,
                 drop=0., drop_path=0., use_layer_scale=True, layer_scale_init_value=1e-5, use_adaptive_fusion=True,
                 num_heads=4, qkv_bias=False, qkv_scale=None, hierarchical=False, num_joints=17,
                 use_temporal_similarity=True, temporal_connection_len=1, use_tcn=False, graph_only=False,
                 neighbour_num=4, n_frames=27, compute_dtype="bf16"):
        super().__init__()
        unsupported = []
        if (dim_in, dim_feat, dim_rep, dim_out, mlp_ratio, num_joints) != (3, 128, 512, 3, 4, 17):
            unsupported.append("dims other than dim_in=3, dim_feat=128, dim_rep=512, dim_out=3, mlp_ratio=4, num_joints=17")
        if act_layer is not nn.GELU:
            unsupported.append("act_layer other than nn.GELU")
        if attn_drop or drop or drop_path:
            unsupported.append("dropout > 0 (every shipped yaml uses 0)")
        if qkv_bias or qkv_scale is not None or hierarchical or not use_temporal_similarity:
            unsupported.append("qkv_bias / qkv_scale / hierarchical / use_temporal_similarity=False")
        if num_heads not in (2, 4, 8, 16):
            unsupported.append("num_heads not in {2, 4, 8, 16} (8 = configs/*.yaml:84 runs the MFMA attention kernels, the others generic ones)")
        if neighbour_num not in (1, 2, 3, 4) or not 4 <= n_frames <= 256:
            unsupported.append("neighbour_num outside 1..4 or n_frames outside [4, 256]")
        if num_heads == 2 and n_frames > 157:
            unsupported.append("num_heads=2 with n_frames > 157 (the generic attention backward keeps a head's track in LDS)")
        if unsupported:
            raise NotImplementedError("kasportsformer_amd builds the shipped configuration only; unsupported: " + "; ".join(unsupported))
        if compute_dtype not in ("bf16", "fp32"):
            raise ValueError("compute_dtype must be 'bf16' or 'fp32'")
        self.n_layers, self.n_frames, self.num_heads, self.compute_dtype = n_layers, n_frames, num_heads, compute_dtype
        self.attach_param_grads = True      # False: gradients stay in self.flat_grad only (FusedAdamW path)
        self.grad_stage_hook = None         # callable(stage, begin, end) after each backward stage (data parallel)
        self.flat_grad = None

        # ---- parameter tree, same names / registration order as the reference ----
        self.joints_embed = nn.Linear(dim_in, dim_feat)
        self.bone_embed = nn.Linear(dim_in, dim_feat)
        self.limb_embed = nn.Linear(dim_in, dim_feat)
        self.pos_embed = nn.Parameter(torch.zeros(1, num_joints, dim_feat))
        self.bone_pos_embed = nn.Parameter(torch.zeros(1, num_joints, dim_feat))
        self.limb_pos_embed = nn.Parameter(torch.zeros(1, num_joints, dim_feat))
        self.norm = nn.LayerNorm(dim_feat)
        self.bone_refusion = _Box()
        groups = []
        for n in _LIMB_SIZES:                  # modules/bone_MLP.py:13-15
            g = _Box()
            g.mlp_dir_x, g.mlp_dir_y, g.mlp_len = _mlp_box(n, 16, 1), _mlp_box(n, 16, 1), _mlp_box(n, 16, 1)
            groups.append(g)
        self.bone_refusion.mlp_layers = nn.Sequential(*groups)
        layers = []
        for _ in range(n_layers):
            layer = _Box()
            for kind in BLOCK_KINDS:
                setattr(layer, kind, _former_box(kind, dim_feat, n_frames, mlp_ratio, layer_scale_init_value, use_layer_scale))
            layer.fusion_three_channel = nn.Linear(dim_feat * 3, 3)
            layer.fusion_three_channel.weight.data.fill_(0)        # model/KASportsFormer.py:264-266
            layer.fusion_three_channel.bias.data.fill_(1 / 3)
            layers.append(layer)
        self.layers_with_bone = nn.Sequential(*layers)
        self.rep_logit = nn.Sequential(OrderedDict([("fc", nn.Linear(dim_feat, dim_rep)), ("act", nn.Tanh())]))
        self.head = nn.Linear(dim_rep, dim_out)

        # ---- native handle + flat storage ----
        self._cfg = _lib.KasfConfig(n_layers, n_frames, num_heads, neighbour_num, 1 if use_adaptive_fusion else 0,
                                    _lib.DTYPE_BF16 if compute_dtype == "bf16" else _lib.DTYPE_F32)
        self._lib = _lib.load()
        self._layout = C.c_void_p()
        _lib.check(self._lib.kasf_model_create_layout_only(C.byref(self._cfg), C.byref(self._layout)))
        self._handle = None                    # device handle, created on first CUDA use
        self._p_entries = {n: (o, s) for n, o, s in _lib.param_entries(self._layout)}
        self._b_entries = {n: (o, s) for n, o, s in _lib.buffer_entries(self._layout)}
        self.n_flat = self._lib.kasf_param_count(self._layout)
        self.n_live = self._lib.kasf_param_live_count(self._layout)
        names = dict(self.named_parameters())
        # use_layer_scale=False: the library's layer-scale slices stay in the flat array as the constant 1 (x + 1 * f(x) is the reference's x + f(x) bit for bit);
        # they are not parameters: absent from state_dict / parameters(), never updated (FusedAdamW puts the 1 back after its flat update)
        self._const_one = sorted((o, int(torch.Size(s).numel())) for n, (o, s) in self._p_entries.items() if n not in names)
        if set(names) - set(self._p_entries) or any(".layer_scale_" not in n for n in set(self._p_entries) - set(names)) or (use_layer_scale and self._const_one):
            raise RuntimeError("native layout and module tree disagree on parameter names")
        for n, p in names.items():
            if tuple(p.shape) != tuple(self._p_entries[n][1]):
                raise RuntimeError(f"shape mismatch for {n}")
        self._packed = None
        self._packed_dirty = True
        self.static_weights = False         # True: trust mark_weights_dirty() instead of re-packing on every forward (inference serving)
        self._flatten()
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.mark_weights_dirty())

    # ------------------------------------------------------------------ flat storage management
    def _flatten(self):
        """(Re)creates the flat fp32 parameter / buffer arrays on the parameters' device and re-points
        every nn.Parameter / buffer at its slice."""
        dev = self.pos_embed.device
        flat = torch.zeros(self.n_flat, dtype=torch.float32, device=dev)
        self._live = []
        for n, p in self.named_parameters():
            off, shape = self._p_entries[n]
            view = flat[off:off + p.numel()].view(shape)
            view.copy_(p.data)
            p.data = view
            if off < self.n_live:
                self._live.append((p, off, p.numel(), tuple(shape)))
        self._const_index = None
        if self._const_one:
            self._const_index = torch.cat([torch.arange(o, o + n, device=dev) for o, n in self._const_one])
            flat.index_fill_(0, self._const_index, 1.0)
        self._flat = flat
        fbuf = torch.zeros(self._lib.kasf_buffer_count(self._layout), dtype=torch.float32, device=dev)
        nbt = []
        for n, b in list(self.named_buffers()):
            if n.endswith("num_batches_tracked"):
                nbt.append((n, b))
                continue
            off, shape = self._b_entries[n]
            view = fbuf[off:off + b.numel()].view(shape)
            view.copy_(b)
            self._set_buffer(n, view)
        self._flat_buffers = fbuf
        counters = torch.zeros(max(1, len(nbt)), dtype=torch.int64, device=dev)
        for i, (n, b) in enumerate(nbt):
            counters[i] = b
            self._set_buffer(n, counters[i])
        self._nbt = counters
        self._packed = None
        self._packed_dirty = True

    def _set_buffer(self, dotted, tensor):
        mod = self
        *path, leaf = dotted.split(".")
        for part in path:
            mod = getattr(mod, part)
        mod._buffers[leaf] = tensor

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flatten()
        return out

    def mark_weights_dirty(self):
        """The kernel-side weight arena is stale: re-pack before the next forward.  Only matters with ``static_weights=True``; by default
        every forward re-packs (one ~0.1 ms launch), because an ``nn.Parameter`` can be edited in ways no hook sees (``torch.optim`` steps,
        ``p.data.fill_()``, ``load_state_dict``) and the reference module picks all of them up on its next call."""
        self._packed_dirty = True

    # ------------------------------------------------------------------ native calls
    def _device_handle(self):
        if self._handle is None:
            h = C.c_void_p()
            _lib.check(self._lib.kasf_model_create(C.byref(self._cfg), C.byref(h)))
            self._handle = h
        return self._handle

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _ensure_packed(self):
        h = self._device_handle()
        if self._packed is None or self._packed.device != self._flat.device:
            self._packed = torch.empty(self._lib.kasf_packed_bytes(h), dtype=torch.uint8, device=self._flat.device)
            self._packed_dirty = True
        if self._packed_dirty or not self.static_weights:
            _lib.check(self._lib.kasf_pack_weights(h, self._flat.data_ptr(), self._packed.data_ptr(), self._stream()))
            self._packed_dirty = False

    def _check_input(self, x):
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[2] != 17 or x.shape[3] != 3:
            raise RuntimeError(f"expected x of shape [B, T, 17, 3], got {tuple(x.shape) if isinstance(x, torch.Tensor) else type(x)}")
        if x.shape[1] != self.n_frames:
            raise RuntimeError(f"running_mean should contain {x.shape[1]} elements not {self.n_frames}: model was built with n_frames={self.n_frames}")
        if not x.is_cuda or not self._flat.is_cuda:
            raise RuntimeError("kasportsformer_amd runs on an MI355X only (move model and input to 'cuda'); there is no CPU fallback")
        if x.dtype != torch.float32:
            raise RuntimeError("x must be float32")

    def _launch_forward(self, x, return_rep, keep):
        h = self._device_handle()
        self._ensure_packed()
        x = x.contiguous()
        B = x.shape[0]
        flags = (_lib.FLAG_TRAIN if self.training else 0) | (_lib.FLAG_RETURN_REP if return_rep else 0) | (_lib.FLAG_KEEP if keep else 0)
        nbytes = self._lib.kasf_workspace_bytes(h, B, flags)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        out = torch.empty((B, self.n_frames, 17, 512 if return_rep else 3), dtype=torch.float32, device=x.device)
        _lib.check(self._lib.kasf_forward(h, self._flat.data_ptr(), self._packed.data_ptr(), self._flat_buffers.data_ptr(), x.data_ptr(),
                                          out.data_ptr(), ws.data_ptr(), nbytes, B, flags, self._stream()))
        if self.training:
            self._nbt += 1                      # BatchNorm num_batches_tracked (graph.py:37)
        return out, ws, flags

    @torch.no_grad()
    def stage_outputs(self, x):
        """The forward pass is ONE library call, so forward hooks on the sub-modules never fire (their `forward` raises).  This is the replacement for
        hook-based inspection: runs the forward in the module's current mode and returns `(out, stages)` where `stages` maps the reference's module names
        to what a forward hook on that module would have seen as its output -- `layers_with_bone.<l>.<kind>` (the six FormerModules of a layer,
        KASportsFormer.py:204-286), `layers_with_bone.<l>` (the gated sum) -- plus `joints_embed` / `bone_embed` / `limb_embed` (the three streams entering
        layer 0, position embeddings added) and `rep_logit`.  Tensors are fp32 copies `[B, T, 17, C]`; bf16 mode returns the stored bf16 values widened.
        Training mode also updates the BatchNorm running statistics once, like any forward."""
        self._check_input(x)
        out, ws, flags = self._launch_forward(x, False, keep=True)
        B, T = x.shape[0], self.n_frames
        ents = _lib.ws_entries(self._device_handle(), B, flags)
        act = torch.bfloat16 if self.compute_dtype == "bf16" else torch.float32
        def get(name, width=128):
            off, numel, kind = ents[name]
            dt = {0: act, 1: torch.float32}[kind]
            nbytes = numel * torch.empty((), dtype=dt).element_size()
            return ws[off:off + nbytes].view(dt).float().view(B, T, 17, width).clone()
        stages = {"joints_embed": get("x_joint"), "bone_embed": get("x_bone"), "limb_embed": get("x_limb")}
        for l in range(self.n_layers):
            for kind in BLOCK_KINDS:
                stages[f"layers_with_bone.{l}.{kind}"] = get(f"L{l}.{kind}.x_out")
            stages[f"layers_with_bone.{l}"] = get(f"L{l}.gate_out")
        stages["rep_logit"] = get("rep", 512)
        return out, stages

    def _launch_backward(self, ws, dout, B, flags):
        h = self._device_handle()
        dout = dout.contiguous().float()
        # Gradient accumulation (a second backward before the optimizer step) follows torch PER PARAMETER: a tensor whose .grad is None (never
        # touched, or zeroed by an optimizer that owns only a subset of the parameters -- e.g. fine-tuning `head` alone) restarts from this
        # pass's gradient, every other one holds the SUM -- in p.grad and in the flat array FusedAdamW reads.  The engine's finishing kernels
        # scale weight gradients in place, so every backward runs into a zeroed array of its own and is added to the running one afterwards
        # (one extra pass over 117 MB, only in the accumulating case).  With attach_param_grads=False the flat array accumulates until
        # zero_grad() of this module or of FusedAdamW.
        restart = []                                   # (offset, numel) of live tensors that restart although others accumulate
        if self.flat_grad is None:
            fresh = True
        elif not self.attach_param_grads:
            fresh = False
        else:
            restart = sorted((off, n) for p, off, n, _ in self._live if p.grad is None)
            fresh = len(restart) == len(self._live)
        if not fresh and self.grad_stage_hook is not None:
            raise NotImplementedError("gradient accumulation with the overlapped (stage-hooked) all-reduce is not built: zero_grad() between backward "
                                      "passes, or DataParallel(overlap=False)")
        g = torch.zeros(self.n_flat, dtype=torch.float32, device=dout.device)
        stages = self._lib.kasf_backward_stages(h)
        args = (h, self._flat.data_ptr(), self._packed.data_ptr(), dout.data_ptr(), g.data_ptr(), ws.data_ptr(), ws.numel(), B, flags)
        if self.grad_stage_hook is None:
            _lib.check(self._lib.kasf_backward(*args, 0, stages, self._stream()))
        else:
            # `grad_stage_group` consecutive stages per call: the gradient ranges of consecutive layers are adjacent, so each call finalises one
            # contiguous bucket (two for the last group: layers + top level) that the hook hands to the all-reduce while the next group runs
            b, e = C.c_int64(), C.c_int64()
            group = max(1, int(getattr(self, "grad_stage_group", 1)))
            for st in range(0, stages, group):
                end = min(st + group, stages)
                _lib.check(self._lib.kasf_backward(*args, st, end, self._stream()))
                spans = []                                           # adjacent stage ranges merge; the top-level range stays on its own
                for k in range(st, end):
                    _lib.check(self._lib.kasf_stage_grad_range(h, k, C.byref(b), C.byref(e)))
                    if e.value > b.value:
                        if spans and spans[-1][0] == e.value:
                            spans[-1][0] = b.value
                        elif spans and spans[-1][1] == b.value:
                            spans[-1][1] = e.value
                        else:
                            spans.append([b.value, e.value])
                for lo, hi in spans:
                    self.grad_stage_hook(st, g[lo:hi])
        if fresh:
            self.flat_grad = g
        else:
            lo = hi = None
            for off, n in restart:                     # zero what restarts (adjacent ranges merged), then add this pass everywhere
                if hi is not None and off <= hi + 3:
                    hi = off + n
                    continue
                if hi is not None:
                    self.flat_grad[lo:hi].zero_()
                lo, hi = off, off + n
            if hi is not None:
                self.flat_grad[lo:hi].zero_()
            self.flat_grad += g
        if self.attach_param_grads:
            total = self.flat_grad
            untouched = (self.head.weight, self.head.bias) if flags & _lib.FLAG_RETURN_REP else ()
            for p, off, n, shape in self._live:
                if p.grad is None:
                    if any(p is q for q in untouched):
                        continue                     # return_rep=True: the head took no part, its .grad stays None as in the reference
                    p.grad = total[off:off + n].view(shape)      # a view of the flat array: later backward passes accumulate into it in place
                elif p.grad.data_ptr() != total.data_ptr() + 4 * off:
                    p.grad += g[off:off + n].view(shape)         # a tensor the caller put there: add this pass's gradient like autograd would

    def zero_grad(self, set_to_none: bool = True):
        """nn.Module.zero_grad, and the flat gradient array with it (it is what FusedAdamW and attach_param_grads=False callers read)."""
        super().zero_grad(set_to_none=set_to_none)
        if set_to_none:
            self.flat_grad = None
        elif self.flat_grad is not None:
            self.flat_grad.zero_()

    # ------------------------------------------------------------------ reference interface
    def forward(self, x, return_rep=False):
        """x [B, T, 17, 3] float32 on the GPU -> [B, T, 17, 3] (or the [B, T, 17, 512] tanh features).
        Returns a fresh writable tensor and never modifies x (train_and_evaluate_sp.py:55 mutates the output)."""
        self._check_input(x)
        if x.shape[0] == 0:        # the reference returns an empty result in both modes (checked against it); no kernel can be launched on zero clips
            return x.new_empty((0, self.n_frames, 17, 512 if return_rep else 3))
        if torch.is_grad_enabled() and self.pos_embed.requires_grad:
            return _KasfFunction.apply(x, self.pos_embed, self, bool(return_rep))
        out, _, _ = self._launch_forward(x, return_rep, keep=False)
        return out

    def __del__(self):
        try:
            for h in (self._handle, self._layout):
                if h is not None and h.value:
                    self._lib.kasf_model_destroy(h)
        except Exception:
            pass


def set_single_stream(on: bool = True) -> None:
    """Process-wide: run the three branches of every layer back to back on the caller's stream instead of on three streams (about 23 % of the training
    throughput since the half-chip MLP launches of round 4: 3,389 against 4,428 clips/s): the mode isolated kernel profiles are taken in.  Not a determinism switch -- gradients are bit-reproducible from run to run either way
    and both settings give the same bits."""
    _lib.load().kasf_set_single_stream(1 if on else 0)


def is_single_stream() -> bool:
    return bool(_lib.load().kasf_get_single_stream())


set_deterministic, is_deterministic = set_single_stream, is_single_stream      # the round-3 names


def set_fused_attention_backward(on=True) -> None:
    """Process-wide, opt-in (round 6; C-ABI: ``kasf_set_fused_attn_bwd``): run the backward of every attention / bone block with groups of at most 32 positions (bf16,
    8 heads) as one launch that re-forms q | k | v and the attention output from the block input -- the training forward then saves none of them (workspace at T = 27,
    B = 256: 26.9 -> 14.4 GB) -- followed by one streaming weight-gradient launch.  Same results at bf16 rounding level, bit-reproducible, 15 % slower per training
    step than the default four-launch sequence on MI355X.  Change it only between steps: the forward and the backward of one step must agree."""
    _lib.load().kasf_set_fused_attn_bwd(int(on))          # True = all block kinds; an int selects kinds by bit (1 self spatial, 2 self temporal, 4 bone spatial, 8 bone temporal)


def is_fused_attention_backward() -> bool:
    return bool(_lib.load().kasf_get_fused_attn_bwd())


def load_model(args) -> nn.Module:
    """model/model_tools.py:79-96: builds the model from yaml-style fields (attribute or mapping access)."""
    get = (lambda k: args[k]) if isinstance(args, dict) else (lambda k: getattr(args, k))
    act_mapper = {"gelu": nn.GELU, "relu": nn.ReLU}
    if get("model_name") != "KASportsFormer":
        raise Exception("Unexpected model name")
    try:
        cd = get("compute_dtype")
    except (KeyError, AttributeError):
        cd = "bf16"
    return KASportsFormer(n_layers=get("n_layers"), dim_in=get("dim_in"), dim_feat=get("dim_feat"), dim_rep=get("dim_rep"), dim_out=get("dim_out"),
                          mlp_ratio=get("mlp_ratio"), act_layer=act_mapper[get("act_layer")], attn_drop=get("attn_drop"), drop=get("drop"),
                          drop_path=get("drop_path"), use_layer_scale=get("use_layer_scale"), layer_scale_init_value=get("layer_scale_init_value"),
                          use_adaptive_fusion=get("use_adaptive_fusion"), num_heads=get("num_heads"), qkv_bias=get("qkv_bias"),
                          qkv_scale=get("qkv_scale"), hierarchical=get("hierarchical"), num_joints=get("num_joints"),
                          use_temporal_similarity=get("use_temporal_similarity"), temporal_connection_len=get("temporal_connection_len"),
                          use_tcn=get("use_tcn"), graph_only=get("graph_only"), neighbour_num=get("neighbour_num"), n_frames=get("n_frames"),
                          compute_dtype=cd)
