"""Build-time guard against COMPILER-GENERATED packed-fp32 arithmetic with operand-select broadcasts.

Round 3 found `k_gcn_bwd2_spatial` / `_temporal` returning values that depended on which other kernels shared the SIMD.  The source there is plain scalar C
(`dy = sc * (r - c1 - (y - mean) * rstd * c2)`); hipcc's SLP vectoriser had turned it into `v_pk_add_f32` / `v_pk_mul_f32` whose scalar operand is broadcast
with `op_sel` / `op_sel_hi` out of a register PAIR of which only one half is ever written (tools/packed_fp32_repro.hip keeps the pattern and the ISA).
The library is built with `-fno-slp-vectorize` since then.  Rounds 1-4 had ONE hand-written packed-fp32 site, the GELU / GELU' chains of the fused MLP kernels;
round 5 moved those to packed FP16 (`gelu_pairs_h`, `gelu_grad_pairs_h`: v_pk_*_f16 runs beside another wave's MFMAs, v_pk_*_f32 runs IN the matrix pipe), so
the library now holds no packed-fp32 arithmetic with operand select at all.

This test takes the compiler flags FROM THE MAKEFILE (so removing the flag there fails here), compiles every product source to gfx950 assembly (no GPU needed) and
fails on any `v_pk_{add,mul,fma}_f32` carrying `op_sel` / `op_sel_hi` in any kernel.  A second check compiles
csrc/k_gcn.hip WITHOUT the flag and expects the pattern to appear in the kernels that showed the defect -- if a compiler update stops producing it, the flag
(and this guard) can be reconsidered instead of being carried forever.
"""
import concurrent.futures as cf
import os
import re
import shlex
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kasportsformer_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

PK = re.compile(r"^\s+(v_pk_(?:add|mul|fma)_f32)\b(.*)$")
# kernels whose packed fp32 is written out by hand in the source (two-float vector types): none since round 5
HAND_WRITTEN = ()


def makefile_vars():
    text = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^CXXFLAGS\s*=\s*(.*)$", text, re.M).group(1)
    srcs = re.search(r"^SRCS\s*=\s*(.*)$", text, re.M).group(1).split()
    flags = flags.replace("$(ARCH)", "gfx950").replace("$(EXTRA)", "")
    return shlex.split(flags), srcs


def to_asm(src, flags, tag):
    out_dir = os.path.join(CSRC, "build", "asm_" + tag)
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, src.replace(".hip", ".s"))
    deps = [os.path.join(CSRC, f) for f in (src, "common.h", "kernels.h", "tile_ops.h", "Makefile")] + [os.path.join(ROOT, "include", "kasf.h")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    flags = [f for f in flags if f not in ("-fPIC",) and not f.startswith("-I")]
    cmd = [HIPCC] + flags + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


def packed_with_select(asm_path):
    """{kernel name: [instruction lines]} of packed fp32 add / mul / fma instructions that carry op_sel or op_sel_hi."""
    lines = open(asm_path).read().split("\n")
    names = [m.group(1) for m in (re.match(r"^\s+\.amdhsa_kernel\s+(\S+)", ln) for ln in lines) if m]
    found = {}
    for name in names:
        start = next(i for i, ln in enumerate(lines) if ln.startswith(name + ":"))
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        hits = [ln.strip() for ln in lines[start:end] if (m := PK.match(ln)) and "op_sel" in m.group(2)]
        if hits:
            found[name] = hits
    return found


@pytest.fixture(scope="module")
def shipped_asm():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    flags, srcs = makefile_vars()
    with cf.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        return dict(zip(srcs, ex.map(lambda s: to_asm(s, flags, "shipped"), srcs)))


def test_no_compiler_generated_packed_fp32_with_operand_select(shipped_asm):
    problems, hand = [], 0
    for src, path in shipped_asm.items():
        for name, hits in packed_with_select(path).items():
            if any(re.search(rx, name) for rx in HAND_WRITTEN):
                hand += len(hits)
                continue
            problems.append(f"{src}: {name}: {len(hits)} packed-fp32 instructions with op_sel / op_sel_hi, e.g. `{hits[0]}`")
    assert hand == 0          # (that the scanner recognises the pattern at all is the negative control's job, below)
    assert not problems, ("compiler-generated packed fp32 with operand select (is -fno-slp-vectorize still in csrc/Makefile?):\n" + "\n".join(problems))


def test_without_the_flag_the_pattern_comes_back_in_the_gcn_backward():
    """The negative control: the same scanner on k_gcn.hip compiled WITHOUT -fno-slp-vectorize must find the pattern in the BatchNorm-backward kernels that
    gave co-residency-dependent results.  (So the guard above is known to fire when the flag is removed from the Makefile.)"""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    flags, _ = makefile_vars()
    assert "-fno-slp-vectorize" in flags, "csrc/Makefile no longer passes -fno-slp-vectorize"
    found = packed_with_select(to_asm("k_gcn.hip", [f for f in flags if f != "-fno-slp-vectorize"], "slp"))
    hit = [n for n in found if "k_gcn_bwd2" in n]
    assert hit, f"no packed fp32 with operand select in k_gcn_bwd2_* without the flag (kernels with the pattern: {sorted(found)[:6]}): the compiler changed, revisit the flag"
