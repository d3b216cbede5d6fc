"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol declared in
include/kasf.h, its parameter layout covers exactly the reference's state_dict, and the host module keeps
the reference's constructor / state_dict contract.  No compute is launched (there is no GPU here)."""
import ctypes as C
import json
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    from kasportsformer_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "kasf.h")).read()
    declared = set(re.findall(r"\b(kasf_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"kasf_ws_name"}                                   # mentioned in a comment only
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/kasf.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes prototype"
    assert lib.kasf_version() == _lib.ABI_VERSION
    # ... and the other way round: nothing with C linkage leaves the library without a declaration in the header (the host module may only use what is declared)
    import shutil
    import subprocess
    nm = shutil.which("nm")
    if nm is not None:
        out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
        exported = set(re.findall(r" T (kasf_[a-z0-9_]+)$", out, flags=re.M))
        assert exported and not exported - declared, sorted(exported - declared)


def test_header_is_plain_c_and_links_against_the_library(tmp_path):
    """The boundary is a C ABI: include/kasf.h must compile as C99 (no C++ in the signatures) and a C program that only includes it must link against the
    library and run the two calls that need no GPU -- what a cgo / JNI / Rust-FFI binding of the reference's host language would do first."""
    import shutil
    import subprocess
    from kasportsformer_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    hdr = os.path.join(ROOT, "include", "kasf.h")
    r = subprocess.run([gcc, "-std=c99", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    src = tmp_path / "probe.c"
    src.write_text('''#include <stdio.h>
#include "kasf.h"
int main(void) {
    kasf_config cfg = {26, 27, 8, 4, 1, KASF_DTYPE_BF16};
    kasf_model* m = 0;
    if (kasf_model_create_layout_only(&cfg, &m) != 0) { printf("create failed: %s\\n", kasf_last_error()); return 2; }
    printf("%d %lld %lld\\n", kasf_version(), (long long)kasf_param_count(m), (long long)kasf_param_live_count(m));
    kasf_model_destroy(m);
    return 0;
}
''')
    exe = tmp_path / "probe"
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run([gcc, "-std=c99", "-Werror=implicit-function-declaration", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L" + libdir, "-l:libkasf_hip.so", "-Wl,-rpath," + libdir,
                        "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    ver, n, live = r.stdout.split()
    assert int(ver) == _lib.ABI_VERSION and int(n) >= 29365668 and int(n) % 4 == 0 and 0 < int(live) < int(n)      # (the flat array is padded to a multiple of 4)


def test_layout_covers_reference_state_dict(golden_dir):
    from kasportsformer_amd import _lib
    lib = _lib.load()
    man = json.load(open(os.path.join(golden_dir, "state_dict_manifest.json")))
    cfg = _lib.KasfConfig(26, 27, 8, 4, 1, 1)
    h = C.c_void_p()
    _lib.check(lib.kasf_model_create_layout_only(C.byref(cfg), C.byref(h)))
    params = {n: (o, s) for n, o, s in _lib.param_entries(h)}
    bufs = {n: (o, s) for n, o, s in _lib.buffer_entries(h)}
    ref = {e[0]: tuple(e[1]) for e in man["entries"]}
    nbt = {k for k in ref if k.endswith("num_batches_tracked")}
    assert set(params) | set(bufs) | nbt == set(ref)
    for n, (o, s) in {**params, **bufs}.items():
        assert tuple(s) == ref[n], n
    # slices do not overlap and are 16-byte aligned
    spans = sorted((o, o + int(torch.Size(s).numel())) for o, s in params.values())
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and all(o % 4 == 0 for o, _ in spans)
    live = lib.kasf_param_live_count(h)
    dead = {n for n, (o, _) in params.items() if o >= live}
    assert dead == set(man["grad_none"]) and len(dead) == 208
    assert sum(int(torch.Size(s).numel()) for _, s in params.values()) == man["n_params"]
    # backward stages finalise disjoint gradient ranges that tile [0, live)
    b, e = C.c_int64(), C.c_int64()
    ranges = []
    for st in range(lib.kasf_backward_stages(h)):
        _lib.check(lib.kasf_stage_grad_range(h, st, C.byref(b), C.byref(e)))
        if e.value > b.value:
            ranges.append((b.value, e.value))
    ranges.sort()
    assert ranges[0][0] == 0 and ranges[-1][1] == live and all(a[1] == c[0] for a, c in zip(ranges, ranges[1:]))
    assert lib.kasf_workspace_bytes(h, 256, 1) > lib.kasf_workspace_bytes(h, 256, 0) > 0
    lib.kasf_model_destroy(h)
    # plain-mean fusion (use_adaptive_fusion=False, KASportsFormer.py:284): the gate Linear never gets a gradient either -> never-updated region
    cfg2 = _lib.KasfConfig(26, 27, 8, 4, 0, 1)
    _lib.check(lib.kasf_model_create_layout_only(C.byref(cfg2), C.byref(h)))
    params2 = {n: (o, s) for n, o, s in _lib.param_entries(h)}
    live2 = lib.kasf_param_live_count(h)
    dead2 = {n for n, (o, _) in params2.items() if o >= live2}
    assert set(params2) == set(params) and dead2 == dead | {n for n in params if ".fusion_three_channel." in n} and len(dead2) == 208 + 52
    lib.kasf_model_destroy(h)


def test_host_module_contract(golden_dir):
    import kasportsformer_amd as K
    man = json.load(open(os.path.join(golden_dir, "state_dict_manifest.json")))
    m = K.KASportsFormer(n_layers=26, dim_in=3, dim_feat=128, dim_rep=512, dim_out=3, mlp_ratio=4, num_heads=8, n_frames=27)
    sd = m.state_dict()
    assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()] == man["entries"]
    assert sum(p.numel() for p in m.parameters()) == 29365668
    # parameters are views of one flat array: an in-place optimizer update is visible to the kernels
    w = m.layers_with_bone[3].att_spatial.mixer.qkv.weight
    off = m._p_entries["layers_with_bone.3.att_spatial.mixer.qkv.weight"][0]
    with torch.no_grad():
        w.add_(1.0)
    assert torch.equal(m._flat[off:off + w.numel()].view_as(w), w.detach())
    # load_state_dict keeps the flat storage
    from oracle import kasf_oracle as O
    fill = O.name_seeded_fill(sd)
    m.load_state_dict(fill, strict=True)
    assert torch.equal(m._flat[off:off + w.numel()].view_as(w), fill["layers_with_bone.3.att_spatial.mixer.qkv.weight"])
    # same default init as the reference's constructor order under the same seed (oracle mirrors it)
    torch.manual_seed(114514)
    a = K.KASportsFormer(n_layers=1, num_heads=8)
    torch.manual_seed(114514)
    b = O.KASportsFormerOracle(n_layers=1, num_heads=8)
    for (n1, p1), (n2, p2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert n1 == n2 and torch.equal(p1, p2), n1


def test_unsupported_configurations_raise_like_the_reference():
    import kasportsformer_amd as K
    from torch import nn
    assert K.KASportsFormer(n_layers=1).num_heads == 4     # the reference's constructor default builds (every yaml overrides it with 8)
    with pytest.raises(NotImplementedError):
        K.KASportsFormer(num_heads=3)                      # 128 is not divisible by it: the reference fails later, in the reshape
    with pytest.raises(NotImplementedError):
        K.KASportsFormer(num_heads=8, n_frames=3)          # torch.topk(k=4) over 3 frames raises in the reference too
    with pytest.raises(NotImplementedError):
        K.KASportsFormer(num_heads=8, drop=0.1)
    with pytest.raises(NotImplementedError):
        K.KASportsFormer(n_layers=1, num_heads=2, n_frames=158)   # the generic attention backward keeps a head's track in LDS: rejected up front, not at the first backward()
    assert K.KASportsFormer(n_layers=1, num_heads=2, n_frames=157).n_frames == 157
    m2 = K.KASportsFormer(n_layers=1, num_heads=8, use_layer_scale=False)          # accepted since round 3: the layer-scale slices become the constant 1, not parameters
    assert not any("layer_scale" in k for k in m2.state_dict()) and len(m2._const_one) == 12 and bool((m2._flat[m2._const_index] == 1.0).all())
    assert K.KASportsFormer(n_layers=1, num_heads=8, neighbour_num=2).n_frames == 27
    with pytest.raises(NotImplementedError):
        K.KASportsFormer(num_heads=8, act_layer=nn.ReLU)
    m = K.KASportsFormer(n_layers=1, num_heads=8, use_tcn=False, graph_only=False, temporal_connection_len=1)   # dead kwargs accepted
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 27, 17, 3))                       # no CPU fallback
    args = dict(model_name="KASportsFormer", n_layers=1, dim_in=3, dim_feat=128, dim_rep=512, dim_out=3, mlp_ratio=4, act_layer="gelu",
                attn_drop=0.0, drop=0.0, drop_path=0.0, use_layer_scale=True, layer_scale_init_value=1e-5, use_adaptive_fusion=True,
                num_heads=8, qkv_bias=False, qkv_scale=None, hierarchical=False, num_joints=17, use_temporal_similarity=True,
                temporal_connection_len=1, use_tcn=False, graph_only=False, neighbour_num=4, n_frames=27)
    assert isinstance(K.load_model(args), K.KASportsFormer)
    with pytest.raises(Exception):
        K.load_model(dict(args, model_name="MotionAGFormer"))


def test_device_free_entry_points_under_address_and_ub_sanitizers(tmp_path):
    """SURVEY section 5.2 (sanitizers; GPU ASan is not available on the pool, so: the host side).  `make asan` builds engine.hip's host code with
    -fsanitize=address,undefined; a C driver built with the same runtime creates layout-only handles for the shipped and for corner configurations and walks
    every query that needs no device -- names with exact-fit and too-short buffers, out-of-range indices, every stage and workspace entry in every flag
    combination -- then destroys them.  Any heap / stack / global overflow, use after free, signed overflow or misaligned access in that code aborts the driver."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not (os.path.exists(hipcc) and os.path.exists(clang)):
        pytest.skip("ROCm clang not available")
    csrc = os.path.join(ROOT, "kasportsformer_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "-j", "4", "all", "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    src = tmp_path / "walk.c"
    src.write_text(r'''#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "kasf.h"
static int walk(kasf_config cfg) {
    kasf_model* m = 0;
    if (kasf_model_create_layout_only(&cfg, &m) != 0) { printf("create failed: %s\n", kasf_last_error()); return 2; }
    long long total = 0;
    int32_t n = kasf_param_entries(m);
    for (int32_t i = -1; i <= n; ++i) {                          /* one past both ends on purpose */
        char* name = malloc(96);                                 /* heap buffers: ASan sees a one-byte overrun */
        int64_t off = 0, shape[4] = {0, 0, 0, 0};
        int32_t nd = 0;
        int rc = kasf_param_entry(m, i, name, 96, &off, &nd, shape);
        if ((i < 0 || i >= n) != (rc != 0)) { printf("param index %d: rc %d\n", i, rc); return 3; }
        if (rc == 0) {
            size_t len = strlen(name);
            char* tight = malloc(len + 1);                       /* exact fit */
            if (kasf_param_entry(m, i, tight, (int32_t)len + 1, &off, &nd, shape) != 0 || strcmp(tight, name) != 0) { printf("exact-fit name %s\n", name); return 4; }
            free(tight);
            char* small = malloc(4);                             /* too short: an error or a truncated, terminated name -- never a write past 4 bytes */
            (void)kasf_param_entry(m, i, small, 4, &off, &nd, shape);
            free(small);
            total += off;
        }
        free(name);
    }
    n = kasf_buffer_entries(m);
    for (int32_t i = -1; i <= n; ++i) {
        char name[96];
        int64_t off = 0, shape[4];
        int32_t nd = 0;
        (void)kasf_buffer_entry(m, i, name, 96, &off, &nd, shape);
    }
    for (int32_t st = -1; st <= kasf_backward_stages(m); ++st) {
        int64_t b = 0, e = 0;
        int rc = kasf_stage_grad_range(m, st, &b, &e);
        if (rc == 0 && (b < 0 || e < b || e > kasf_param_count(m))) { printf("stage %d: [%lld, %lld)\n", st, (long long)b, (long long)e); return 5; }
    }
    const int32_t batches[4] = {0, 1, 7, 256};
    for (int bi = 0; bi < 4; ++bi)
        for (int32_t flags = 0; flags < 8; ++flags) {
            int64_t bytes = kasf_workspace_bytes(m, batches[bi], flags);
            int32_t ne = kasf_ws_entries(m, batches[bi], flags);
            for (int32_t i = -1; i <= ne; ++i) {
                char* name = malloc(64);
                int64_t off = 0, numel = 0;
                int32_t kind = 0;
                int rc = kasf_ws_entry(m, batches[bi], flags, i, name, 64, &off, &numel, &kind);
                if (rc == 0 && (off < 0 || off > bytes)) { printf("ws entry %s outside the workspace\n", name); return 6; }
                free(name);
            }
        }
    total += kasf_packed_bytes(m) + kasf_param_live_count(m) + kasf_buffer_count(m);
    kasf_model_destroy(m);
    kasf_model_destroy(0);
    return total > 0 ? 0 : 7;
}
int main(void) {
    kasf_config shipped = {26, 27, 8, 4, 1, KASF_DTYPE_BF16};
    kasf_config long_clip = {1, 256, 8, 4, 1, KASF_DTYPE_F32};
    kasf_config short_clip = {2, 4, 4, 1, 0, KASF_DTYPE_BF16};
    kasf_config t81 = {26, 81, 8, 4, 1, KASF_DTYPE_BF16};
    int rc;
    if ((rc = walk(shipped)) || (rc = walk(long_clip)) || (rc = walk(short_clip)) || (rc = walk(t81))) return rc;
    kasf_model* m = 0;
    (void)kasf_model_create_layout_only(0, &m);                  /* null arguments are errors, not crashes */
    (void)kasf_model_create_layout_only(&shipped, 0);
    printf("ok %s\n", kasf_last_error());
    return 0;
}
''')
    exe = tmp_path / "walk"
    libdir = os.path.join(ROOT, "kasportsformer_amd")
    r = subprocess.run([clang, "-std=c99", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared-libsan", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                        "-L" + libdir, "-l:libkasf_hip_asan.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib/llvm/lib/clang/22/lib/linux"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.stdout[-2000:], r.stderr[-4000:])
