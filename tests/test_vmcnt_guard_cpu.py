"""Build-time guard for the hand-counted ``s_waitcnt vmcnt(N)`` waits (guard the class, not the instance).

The persistent kernels keep LDS-direct (``global_load_lds_dwordx4``) loads in flight across tiles and wait with ``vmcnt(N)`` for "everything
but my N youngest vector-memory operations".  That is only correct if those N youngest operations are exactly what the source assumes.  hipcc is
free to move ordinary loads and stores around the inline-asm statements, to split a store, or to spill (a spill is a scratch access: one more
entry in the queue) -- round 1 shipped a cold-start-only race because hipcc had sunk weight loads below a counted prologue wait.

This test compiles every source that contains a counted wait to gfx950 assembly (no GPU needed) and walks BACKWARDS from each hand-written wait
(they are recognisable: emitted between ``;;#ASMSTART`` / ``;;#ASMEND``), around the loop's back edge if it sits in a loop, classifying the
vector-memory instructions it meets:
  * default rule: the N youngest are all LDS-direct loads (the look-ahead requests the wait is meant to leave in flight);
  * kernels whose youngest operations may include something else (the fused MLP backward's first iterations) declare their exact youngest-first patterns below;
  * no kernel with counted waits may spill inside a loop (scratch_* between a loop header and its back edge).
"""
import concurrent.futures as cf
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kasportsformer_amd", "csrc")
SOURCES = ["k_mlp3.hip", "k_mlp2.hip", "k_gemm2.hip", "k_gemm.hip", "k_attn_blk.hip"]
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

VMEM = re.compile(r"^\s+(global_load_lds_\w+|global_load_\w+|global_store_\w+|global_atomic_\w+|buffer_load_\w+|buffer_store_\w+|buffer_atomic_\w+|scratch_\w+|flat_\w+)\b")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
BRANCH = re.compile(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)|^\s+s_branch\s+(\.LBB\d+_\d+)")

# youngest-first patterns that are NOT "all LDS-direct loads": (kernel-name regex, N) -> accepted lists of kinds
G4, S4 = ["glds"] * 4, ["store"] * 4
DECLARED = {
    # k_mlp_bwd_s PRODUCER waves (round 5: they issue the look-ahead loads; the consumers' queue holds stores only and is never waited on): the requests of
    # tiles t+3 and t+2 stay in flight.  First iterations (hipcc peels them): the prologue's weight loads (plain loads) may have been sunk below the prologue's
    # uncounted wait -- "all but the two youngest load groups" holds in that form too
    (r"k_mlp_bwd_sE", 8): [G4 + G4, G4 + ["load"] * 4],
}
PATTERN_EXEMPT = ()
SPILL_OUTSIDE_LOOP_OK = ()


def kind(instr: str) -> str:
    if instr.startswith("global_load_lds"):
        return "glds"
    if instr.startswith("scratch"):
        return "scratch"
    if "atomic" in instr:
        return "atomic"
    if "store" in instr:
        return "store"
    return "load"


def compile_to_asm(src: str) -> str:
    out_dir = os.path.join(CSRC, "build", "asm")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, src.replace(".hip", ".s"))
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in ("common.h", "kernels.h", "tile_ops.h")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


def kernels_of(asm_path):
    """[(name, lines)] for every kernel body in the file."""
    lines = open(asm_path).read().split("\n")
    names = [m.group(1) for m in (re.match(r"^\s+\.amdhsa_kernel\s+(\S+)", ln) for ln in lines) if m]
    out = []
    for name in names:
        start = next(i for i, ln in enumerate(lines) if ln.startswith(name + ":"))
        fe = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        out.append((name, lines[start:fe]))
    return out


def loops_of(body):
    """[(header_index, back_edge_index)] of the natural loops that are laid out contiguously: a label and the last later branch back to it."""
    pos = {m.group(1): i for i, ln in enumerate(body) if (m := LABEL.match(ln))}
    loops = {}
    for i, ln in enumerate(body):
        m = BRANCH.match(ln)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in pos and pos[tgt] < i:
                loops[pos[tgt]] = max(loops.get(pos[tgt], -1), i)
    return sorted(loops.items())


def counted_waits(body):
    """Indices and N of the hand-written waits: `s_waitcnt vmcnt(N)` inside an ;;#ASMSTART block."""
    res, in_asm = [], False
    for i, ln in enumerate(body):
        if "#ASMSTART" in ln:
            in_asm = True
        elif "#ASMEND" in ln:
            in_asm = False
        elif in_asm:
            m = re.match(r"^\s+s_waitcnt vmcnt\((\d+)\)", ln)
            if m:
                res.append((i, int(m.group(1))))
    return res


def youngest(body, at, n, loops):
    """Kinds of the n youngest VMEM instructions textually before body[at], wrapping around the innermost contiguous loop that contains it."""
    inner = None
    for h, e in loops:
        if h <= at <= e and (inner is None or h >= inner[0]):
            inner = (h, e)
    order = list(range(at - 1, (inner[0] if inner else 0) - 1, -1))
    if inner:
        order += list(range(inner[1], inner[0] - 1, -1)) * 4      # previous iterations: whole laps from the back edge up to the header
    kinds = []
    for i in order:
        m = VMEM.match(body[i])
        if m:
            kinds.append(kind(m.group(1)))
            if len(kinds) == n:
                break
    return kinds


@pytest.fixture(scope="module")
def asm_files():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    with cf.ThreadPoolExecutor(max_workers=min(5, os.cpu_count() or 1)) as ex:
        return dict(zip(SOURCES, ex.map(compile_to_asm, SOURCES)))


def test_counted_waits_leave_exactly_the_assumed_operations_in_flight(asm_files):
    checked, problems = 0, []
    for src, path in asm_files.items():
        for name, body in kernels_of(path):
            waits = [(i, n) for i, n in counted_waits(body) if n > 0]
            if not waits:
                continue
            loops = loops_of(body)
            for at, n in waits:
                got = youngest(body, at, n, loops)
                if any(re.search(rx, name) for rx in PATTERN_EXEMPT):
                    continue
                want = next((pats for (rx, nn), pats in DECLARED.items() if nn == n and re.search(rx, name)), [["glds"] * n])
                checked += 1
                if len(got) < n:
                    problems.append(f"{src}:{name}: vmcnt({n}) but only {len(got)} vector-memory instructions precede it")
                elif got not in want:
                    problems.append(f"{src}:{name}: vmcnt({n}): youngest-first {got} is none of the assumed {want}")
    assert checked >= 20, f"only {checked} counted waits found: the recogniser is broken"
    assert not problems, "\n".join(problems)


def test_no_spills_inside_loops_of_kernels_with_counted_waits(asm_files):
    problems = []
    for src, path in asm_files.items():
        for name, body in kernels_of(path):
            if not counted_waits(body):
                continue
            loops = loops_of(body)
            for i, ln in enumerate(body):
                if re.match(r"^\s+scratch_", ln):
                    inside = any(h <= i <= e for h, e in loops)
                    if inside or not any(re.search(rx, name) for rx in SPILL_OUTSIDE_LOOP_OK):
                        problems.append(f"{src}:{name}: spill at line {i} ({'inside' if inside else 'outside'} a loop): {ln.strip()}")
    assert not problems, "\n".join(problems[:20])


def _compiler_drains(asm_files):
    """Round 5 (k_mlp_bwd_s, 155 -> 188 us): hipcc places its OWN `s_waitcnt vmcnt(k)` in front of the first use of a value loaded before a loop -- inside the
    loop when nothing uses the value earlier, and with a small k, because it does not see the inline-asm LDS-direct requests issued since.  Every iteration then
    drains the look-ahead.  In a loop that holds a hand-counted wait vmcnt(N), a compiler-generated wait with k < N is that defect (the cure: a register use of
    every pre-loop load right after the prologue's wait -- touch_loaded() in csrc/common.h)."""
    problems = []
    for src, path in asm_files.items():
        for name, body in kernels_of(path):
            hand = counted_waits(body)
            if not hand:
                continue
            hand_at = {i for i, _ in hand}
            all_loops = [(h, e) for h, e in loops_of(body) if "Loop Header" in body[h] or "in Loop:" in body[h]]      # real loops only (hipcc marks their headers); a backward branch into
                                                                                             # a shared block is not one
            innermost = set()                    # every real loop around a hand-counted wait (hipcc lays a loop out as several label-to-branch cycles: the
            for at, n in hand:                   # latch blocks in front of the header, a short cycle for the iterations that skip the body)
                if n > 0:
                    innermost.update((h, e) for h, e in all_loops if h <= at <= e)
            for h, e in sorted(innermost):
                mine = [n for i, n in hand if h <= i <= e and n > 0]
                if not mine or not any("global_load_lds" in body[i] for i in range(h, e + 1)):
                    continue
                in_asm = False
                for i in range(h, e + 1):
                    ln = body[i]
                    if "#ASMSTART" in ln:
                        in_asm = True
                    elif "#ASMEND" in ln:
                        in_asm = False
                    elif not in_asm and i not in hand_at:
                        m = re.match(r"^\s+s_waitcnt\b.*vmcnt\((\d+)\)", ln)
                        if m and int(m.group(1)) < min(mine):
                            problems.append(f"{src}:{name}: compiler-generated '{ln.strip()}' at line {i} inside the loop whose counted wait is vmcnt({min(mine)})")
    return problems


def test_no_compiler_wait_drains_the_look_ahead_inside_a_loop(asm_files):
    problems = _compiler_drains(asm_files)
    assert not problems, "\n".join(problems[:20])


def test_the_drain_guard_sees_the_defect_it_was_written_for():
    """Negative control: k_mlp3.hip with touch_loaded() compiled out must show hipcc's in-loop vmcnt(1) / vmcnt(0) in k_mlp_bwd_s again."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = os.path.join(CSRC, "build", "asm", "k_mlp3_no_touch.s")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-DKASF_NO_TOUCH_LOADED", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, "--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, "k_mlp3.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    problems = _compiler_drains({"k_mlp3.hip": out})
    assert any("k_mlp_bwd_s" in p for p in problems), "the guard no longer recognises the in-loop drain"
