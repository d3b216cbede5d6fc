"""CPU study (test infrastructure, not product): which bf16 roundings cost the 26-layer parity?

Emulates on the CPU oracle the rounding points of the HIP path's bf16 mode, separately:
  S  = the forward residual stream (x_mid, x_out, gate output, embeddings) stored as bf16
  GS = the gradient stream at the same points stored as bf16
  O  = GEMM operands (Linear inputs / weights / outputs and their gradients) rounded to bf16
and prints forward error (max |d| / max(1, max |ref|)) and flat-gradient cosine against the fp32 oracle, with the temporal
neighbour decisions of the fp32 run forced in every variant (tests/gpu_util.py::forced_adjacency says why).

Run:  python tests/studies/mixed_precision_study.py [n_layers]
Result (26 layers, B = 2, seed 5) is quoted in DESIGN.md section 10."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import kasf_oracle as O  # noqa: E402

FLAGS = {"S": False, "GS": False, "O": False}


class _R(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, b):
        ctx.b = b
        return x.bfloat16().float() if f else x

    @staticmethod
    def backward(ctx, g):
        return (g.bfloat16().float() if ctx.b else g), None, None


def stream(x):
    return _R.apply(x, FLAGS["S"], FLAGS["GS"])


def former_forward(self, x, x_limb=None):
    if self.mixer_type == "bone":
        m = self.mixer(self.norm1(x), self.norm1_limb(x_limb))
    else:
        m = self.mixer(self.norm1(x))
    x = stream(x + self.layer_scale_1 * m)
    return stream(x + self.layer_scale_2 * self.mlp(self.norm2(x)))


def layer_forward(self, x, x_bone=None, x_limb=None):
    xa = self.att_temporal(self.att_spatial(x))
    xg = self.graph_temporal(self.graph_spatial(x))
    xb = self.bone_temporal(self.bone_spatial(x if x_bone is None else x_bone, x_limb), x_limb)
    a = self.fusion_three_channel(torch.cat((xa, xg, xb), dim=-1)).softmax(dim=-1)
    return stream(xa * a[..., 0:1] + xg * a[..., 1:2] + xb * a[..., 2:3])


def model_forward(self, x, return_rep=False):
    x_bone = stream(self.bone_embed(O.bone_decompose(x)) + self.bone_pos_embed)
    x_limb = stream(self.limb_embed(self.bone_refusion(x)) + self.limb_pos_embed)
    x = stream(self.joints_embed(x) + self.pos_embed)
    for i, layer in enumerate(self.layers_with_bone):
        x = layer(x, x_bone if i == 0 else None, x_limb)
    x = self.rep_logit(self.norm(x))
    return x if return_rep else self.head(x)


O._FormerModule.forward = former_forward
O._Layer.forward = layer_forward
O.KASportsFormerOracle.forward = model_forward


def add_operand_hooks(model):
    for name, mod in model.layers_with_bone.named_modules():
        if isinstance(mod, torch.nn.Linear) and "fusion" not in name:
            mod.register_forward_pre_hook(lambda m, a: (_R.apply(a[0], FLAGS["O"], FLAGS["O"]),))
            mod.register_forward_hook(lambda m, a, o: _R.apply(o, FLAGS["O"], FLAGS["O"]))


def run(model, x, y, masks, record):
    calls = [0]

    def fn(g, k):
        if record:
            sim = g.detach() @ g.detach().transpose(1, 2)
            thr = sim.topk(k=k, dim=-1)[0][..., -1:]
            masks.append(sim >= thr)
            return masks[-1].to(g.dtype)
        calls[0] += 1
        return masks[calls[0] - 1].to(g.dtype)

    orig = O.temporal_topk_adjacency
    O.temporal_topk_adjacency = fn
    try:
        model.zero_grad()
        pred = model(x)
        loss, _ = O.loss_total(pred, y)
        loss.backward()
    finally:
        O.temporal_topk_adjacency = orig
    return pred.detach(), torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None]).double()


def emulate_samples(samples, L=26):
    """Round 4: today's bf16 mode (O + S + GS) EMULATED on the CPU oracle for each (input seed, weight salt) sample of
    tests/test_gpu_model.py::test_full_depth_26_layers_against_oracle -- what bf16 arithmetic alone does to each sample, no HIP kernel involved.
    Prints one JSON list (committed as tests/golden/bf16_emulation_26layers.json; the GPU test judges the HIP path against it sample by sample)."""
    import json
    out = []
    for seed, salt in samples:
        torch.manual_seed(0)
        model = O.KASportsFormerOracle(n_layers=L, num_heads=8, n_frames=27)
        sd = O.name_seeded_fill(model.state_dict(), salt)
        model.load_state_dict(sd)
        model.train()
        add_operand_hooks(model)
        x, y = O.synthetic_clips(2, 27, seed=seed)
        masks = []
        for k in FLAGS:
            FLAGS[k] = False
        ref, gref = run(model, x, y, masks, True)
        sd_bf = {k: (v.bfloat16().float() if (v.dtype == torch.float32 and v.dim() == 2 and "layers_with_bone" in k and "fusion" not in k) else v) for k, v in sd.items()}
        for k in FLAGS:
            FLAGS[k] = True
        model.load_state_dict(sd_bf)
        pred, g = run(model, x, y, masks, False)
        err = float((pred - ref).abs().max() / max(1.0, float(ref.abs().max())))
        cos = float((g * gref).sum() / (g.norm() * gref.norm()))
        out.append({"seed": seed, "salt": salt, "forward_rel_err": err, "gradient_cosine": cos})
        print(f"# emulated bf16 mode, input seed {seed}, weight salt {salt}: forward err {err:.3e}, gradient cosine {cos:.4f}", file=sys.stderr, flush=True)
    print(json.dumps({"what": "O+S+GS bf16 emulation on the CPU oracle (tests/studies/mixed_precision_study.py samples), 26 layers, B = 2, fp32 run's neighbour decisions forced",
                      "samples": out}, indent=1))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "samples":
        return emulate_samples([(5, 0), (6, 1001), (7, 2002), (8, 3003), (9, 4004), (10, 5005)])
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    torch.manual_seed(0)
    model = O.KASportsFormerOracle(n_layers=L, num_heads=8, n_frames=27)
    sd = O.name_seeded_fill(model.state_dict())
    model.load_state_dict(sd)
    model.train()
    add_operand_hooks(model)
    x, y = O.synthetic_clips(2, 27, seed=5)
    masks = []
    ref, gref = run(model, x, y, masks, True)
    sd_bf = {k: (v.bfloat16().float() if (v.dtype == torch.float32 and v.dim() == 2 and "layers_with_bone" in k and "fusion" not in k) else v) for k, v in sd.items()}
    for label, flags, weights in (("fp32 (self check)", {}, sd), ("S only", {"S": 1}, sd), ("GS only", {"GS": 1}, sd), ("S+GS", {"S": 1, "GS": 1}, sd),
                                  ("O only (operands+weights)", {"O": 1}, sd_bf), ("O+GS  [fp32 fwd stream]", {"O": 1, "GS": 1}, sd_bf),
                                  ("O+S   [fp32 grad stream]", {"O": 1, "S": 1}, sd_bf), ("O+S+GS [today's bf16 mode]", {"O": 1, "S": 1, "GS": 1}, sd_bf)):
        for k in FLAGS:
            FLAGS[k] = bool(flags.get(k))
        model.load_state_dict(weights)
        pred, g = run(model, x, y, masks, False)
        err = float((pred - ref).abs().max() / max(1.0, float(ref.abs().max())))
        cos = float((g * gref).sum() / (g.norm() * gref.norm()))
        print(f"{label:32s} forward err {err:.3e}   gradient cosine {cos:.6f}", flush=True)


if __name__ == "__main__":
    main()
