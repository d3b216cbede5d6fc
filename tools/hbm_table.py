"""Achieved HBM bandwidth of the memory-bound kernels against their ALGORITHMIC bytes, from a rocprofv3 kernel trace.

    python tools/hbm_table.py <single-stream kernel_stats.csv> [B T [three-stream kernel_stats.csv]] > profiles/r2_op_hbm.json

Two byte figures per kernel, kept apart: `algorithmic` = SURVEY §8(d)'s rule -- the external inputs and outputs of the FUSED OP the kernel
belongs to -- and `design` = bytes this design adds on top (the four bf16 dA partials the MLP backward exchanges between its two launches, the saved
activations a kernel writes for the backward pass, LN(x) handed between kernels).  `achieved_GBps` prices the kernel's whole traffic (algorithmic +
design: what it actually has to move), `algorithmic_GBps` only the §8(d) bytes: the second is the roofline figure.
Duration = the AVERAGE over the launches of a trace taken with KASF_SINGLE_STREAM=1 (tools/prof27.sh: the three branches serialised, every kernel
alone on the chip: isolated durations).  If the three-stream trace of the same workload is given, its averages are listed beside them (what a
launch takes while two other kernels share the chip).  Bytes = the external inputs and outputs of the fused op per token x M tokens (s = 2 bytes in
bf16 mode), as listed in DESIGN §4: re-reads from L2 are not counted, so the figure is a lower bound of what moved; peak = 8 TB/s
(MI355X_MICROARCH.md).
"""
import csv, json, re, sys

stats = sys.argv[1]
B, T = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 27)
stats3 = sys.argv[4] if len(sys.argv) > 4 else None
M, s = B * T * 17, 2
# bytes per token (bf16): reads + writes of the kernel's external operands
ALG = {
    "k_linear_r<3, true, false>": ("LN + QKV linear (N=384), writes LN(x)? no", 128 * s + 384 * s),
    "k_linear_r<2, true, false>": ("LN + U|V / KV linear (N=256) + LN(x) out", 128 * s + 256 * s + 128 * s),
    "k_linear_r<1, true, false>": ("LN + Q linear (N=128)", 128 * s + 128 * s),
    "k_linear_r<1, false, true>": ("proj + layer-scale + residual", 3 * 128 * s),
    # round 4: the bf16 engine runs the WG forms (weight gradient inside the data-gradient kernel: no LN(x) output, one bf16 partial dW tile per workgroup as design bytes)
    "k_dgrad_r<3, true, false, false, true, 3, true, false>": ("dqkv.W + LN backward + residual + dW_qkv (dqkv read once)", (384 + 3 * 128) * s, 256 * 384 * 128 * 2 / M),
    "k_dgrad_r<2, true, true, false, true, 3, true, true>": ("GCN: duv.W + LN backward + residual + direct LN(x) gradient + dW_uv, db_uv", (256 + 4 * 128) * s, 256 * 256 * 128 * 2 / M),
    "k_dgrad_r<2, false, false, true, true, 3, true, false>": ("bone: dkv.W + LN backward, accumulates into g_limb, + dW_kv", (256 + 3 * 128) * s, 256 * 256 * 128 * 2 / M),
    "k_dgrad_r<1, true, false, false, true, 3, true, false>": ("bone: dq.W + LN backward + residual + dW_q", (128 + 3 * 128) * s, 256 * 128 * 128 * 2 / M),
    "k_dgrad_r<3, true, false, false, true, 3, false, false>": ("dqkv.W + LN backward + residual, emits LN(x)", (384 + 4 * 128) * s),
    "k_dgrad_r<2, true, true, false, false, 3, false, false>": ("GCN: duv.W + LN backward + residual + direct LN(x) gradient", (256 + 4 * 128) * s),
    "k_dgrad_r<2, false, false, true, true, 3, false, false>": ("bone: dkv.W + LN backward, accumulates into g_limb, emits LN(x_limb)", (256 + 4 * 128) * s),
    "k_dgrad_r<1, true, false, false, true, 3, false, false>": ("bone: dq.W + LN backward + residual, emits LN(x)", (128 + 4 * 128) * s),
    # the fused MLP backward is ONE op by SURVEY §8(d): x_mid, g in, g_in out = 3 x 128 x s per token for the pair of launches; the second launch gets the
    # LayerNorm-backward share (x_mid, g in; g_in out), the first the LN(x) it streams.  Everything else both move is the design's own exchange.
    "k_lnbwd_sum4": ("sum of 4 dA partials + LN backward + residual (+ the weight-gradient partial tiles)", 3 * 128 * s, 4 * 128 * s + 2 * 64 * 65536 * 4 / M),
    "k_gcn_agg_spatial": ("skeleton aggregate: U|V in, y out", (256 + 128) * s),
    "k_gcn_agg_temporal": ("top-4 similarity aggregate: U|V, LN(x) in, y + masks out", (256 + 128 + 128) * s + 12),
    "k_gcn_apply": ("BatchNorm + ReLU + layer-scale + residual: x, LN(x), y in, x_mid out", 4 * 128 * s),
    "k_gcn_bwd1": ("through layer-scale / ReLU: g, LN(x), y in, r out", 4 * 128 * s),
    "k_gcn_bwd2_spatial": ("BatchNorm backward + transposed aggregate: r, y in, dU|dV out", (2 * 128 + 256) * s),
    "k_gcn_bwd2_temporal": ("BatchNorm backward + transposed masked aggregate: r, y, masks in, dU|dV out", (2 * 128 + 256) * s + 12),
    "k_gate_fwd": ("3-way softmax gate: 3 streams in, 1 out (+alpha)", 4 * 128 * s + 16),
    "k_gate_bwd": ("gate backward: g (+2 addends), 3 streams, alpha in; 3 gradients out", 9 * 128 * s + 16),
    "k_attn_blk_fwd_rp<false>": ("fused attention block fwd: x in, x_mid out; q|k|v, o saved for the backward", 2 * 128 * s, (384 + 128) * s),
    "k_attn_blk_fwd_rp<true>": ("fused bone block fwd: x, x_limb in, x_mid out; q, k|v, o saved for the backward", 3 * 128 * s, (384 + 128) * s),
    "k_attn_blk_fwd_rp3<false>": ("fused attention block fwd, 33..96-position groups: x in, x_mid out; q|k|v, o saved", 2 * 128 * s, (384 + 128) * s),
    "k_attn_blk_fwd_rp3<true>": ("fused bone block fwd, 33..96-position groups: x, x_limb in, x_mid out; q, k|v, o saved", 3 * 128 * s, (384 + 128) * s),
    "k_attn_bwd_pers<9>": ("attention backward cores + d_o, 17-position groups: q|k|v, g_mid in; dq|dk|dv out", (384 + 128 + 384) * s),
    "k_attn_bwd_pers<16>": ("attention backward cores + d_o, groups of 18..32 positions: q|k|v, g_mid in; dq|dk|dv out", (384 + 128 + 384) * s),
    "k_attn_bwd_long<3": ("attention backward cores + d_o, 33..96-position groups: q|k|v, g_mid in; dq|dk|dv out", (384 + 128 + 384) * s),
    "k_attn_bwd_kt<": ("attention backward cores + d_o, 33..96-position groups (key-tile-outer form): q|k|v, o, g_mid in; dq|dk|dv out", (384 + 128 + 128 + 384) * s),
    "k_attn_fwd_mfma<3>": ("attention forward core, 33..96-position groups: q|k|v in, o out", (384 + 128) * s),
    "k_wgrad_ring_jobs": ("the proj weight gradient of an attention / bone block: g_mid, o streamed once (qkv / q / kv ride in k_dgrad_r since round 4)", (128 + 128) * s),
    "k_mlp_fwd_s": ("fused MLP fwd: x in, x_out out; LN(x) saved for the backward", 2 * 128 * s, 128 * s),
    "k_mlp_bwd_s": ("fused MLP bwd: LN(x), g in; 4 dA partials out (its x_mid / g_in traffic is booked on k_lnbwd_sum4)", 0, (2 * 128 + 4 * 128) * s),
}
def load(path):
    rows = {}
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", r["Name"])
        if m:
            rows[m.group(1)] = r
    return rows


rows, rows3 = load(stats), (load(stats3) if stats3 else {})
out = {"workload": f"B={B}, T={T}: M = {M} tokens, bf16 training step; durations from a single-stream trace (isolated launches)", "peak_GBps": 8000, "kernels": {}}
for k, spec in ALG.items():
    what, bpt, design = (spec + (0,))[:3]
    hit = next((v for n, v in rows.items() if n.startswith(k)), None)
    if hit is None:
        continue
    nbytes, allbytes, tmin, tavg = bpt * M, (bpt + design) * M, float(hit["MinNs"]) * 1e-9, float(hit["AverageNs"]) * 1e-9
    e = {"what": what, "algorithmic_bytes_per_token": round(bpt, 1), "design_bytes_per_token": round(design, 1), "algorithmic_MB_per_launch": round(nbytes / 1e6, 1),
         "launches_per_step": round(int(hit["Calls"]) / 3, 1), "isolated_avg_us": round(tavg * 1e6, 1), "isolated_min_us": round(tmin * 1e6, 1),
         "algorithmic_GBps": round(nbytes / tavg / 1e9), "algorithmic_frac_of_8TBps": round(nbytes / tavg / 8e12, 3),
         "achieved_GBps": round(allbytes / tavg / 1e9), "frac_of_8TBps": round(allbytes / tavg / 8e12, 3)}
    h3 = next((v for n, v in rows3.items() if n.startswith(k)), None)
    if h3 is not None:
        e["three_stream_avg_us"] = round(float(h3["AverageNs"]) * 1e-3, 1)
    out["kernels"][k] = e
# the MLP backward as the one fused op SURVEY §8(d) prices: 3 x 128 x s bytes per token over BOTH launches
a, b = out["kernels"].get("k_mlp_bwd_s"), out["kernels"].get("k_lnbwd_sum4")
if a and b:
    t = (a["isolated_avg_us"] + b["isolated_avg_us"]) * 1e-6
    out["mlp_backward_chain"] = {"algorithmic_bytes_per_token": 3 * 128 * s, "isolated_avg_us": round(t * 1e6, 1), "algorithmic_GBps": round(3 * 128 * s * M / t / 1e9),
                                 "moved_bytes_per_token": a["design_bytes_per_token"] + b["design_bytes_per_token"] + 3 * 128 * s}
print(json.dumps(out, indent=1))
