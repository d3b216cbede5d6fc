"""Evidence stamps: every profiles/r5_* file says which BUILD it was measured on, and bench.py refuses to quote a file whose stamp
does not match the library it is about to time.

The identity of a build is the sha256 over the sources the library is made of (csrc/*.hip, csrc/*.h, csrc/Makefile, include/kasf.h): it exists on the GPU box
(git does not: the snapshot has no .git), it changes exactly when a kernel changes, and it does not depend on hipcc being bit-reproducible.  The git revision
(from .git_rev, written before the box pass) and the sha256 of the built libkasf_hip.so ride along as information.

    python tools/stamp.py                       # print the stamp of this tree
    python tools/stamp.py --embed a.json b.json # add / refresh "stamp" inside JSON files (a .jsonl file gets one stamp line appended)
    python tools/stamp.py --sidecar DIR PREFIX  # write DIR/PREFIX_STAMP.json covering every DIR/PREFIX_* file that cannot carry a stamp itself (csv, txt)
"""
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_files(root=ROOT):
    c = os.path.join(root, "kasportsformer_amd", "csrc")
    return sorted(glob.glob(os.path.join(c, "*.hip")) + glob.glob(os.path.join(c, "*.h")) + [os.path.join(c, "Makefile"), os.path.join(root, "include", "kasf.h")])


def source_sha256(root=ROOT):
    h = hashlib.sha256()
    for f in source_files(root):
        h.update(os.path.relpath(f, root).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    return h.hexdigest()


def stamp(root=ROOT):
    out = {"source_sha256": source_sha256(root)}
    lib = os.path.join(root, "kasportsformer_amd", "libkasf_hip.so")
    if os.path.exists(lib):
        out["lib_sha256"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()
    rev = os.path.join(root, ".git_rev")
    if os.path.exists(rev):
        out["git_rev"] = open(rev).read().strip()
    return out


def embed(path, st):
    if path.endswith(".jsonl"):
        lines = [l for l in open(path).read().splitlines() if l.strip() and '"stamp"' not in l[:12]]
        open(path, "w").write("\n".join(lines + [json.dumps({"stamp": st})]) + "\n")
        return
    j = json.load(open(path))
    j["stamp"] = st
    json.dump(j, open(path, "w"), indent=1)


def stamp_of(path):
    """The stamp a profile file carries: embedded ("stamp" key / last line of a .jsonl) or through the sidecar <prefix>_STAMP.json next to it; None if neither."""
    try:
        if path.endswith(".json"):
            return json.load(open(path)).get("stamp")
        if path.endswith(".jsonl"):
            for l in reversed(open(path).read().splitlines()):
                if l.strip():
                    j = json.loads(l)
                    return j.get("stamp") if set(j) == {"stamp"} else None
        base = os.path.basename(path)
        side = os.path.join(os.path.dirname(path), base.split("_", 1)[0] + "_STAMP.json")
        j = json.load(open(side))
        return j["stamp"] if base in j.get("files", []) else None
    except (OSError, ValueError, KeyError):
        return None


def is_fresh(path, root=ROOT):
    st = stamp_of(path)
    return bool(st) and st.get("source_sha256") == source_sha256(root)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--embed":
        st = stamp()
        for f in a[1:]:
            embed(f, st)
    elif a and a[0] == "--sidecar":
        d, prefix = a[1], a[2]
        files = sorted(os.path.basename(f) for f in glob.glob(os.path.join(d, prefix + "_*")) if not f.endswith((".json", ".jsonl")))
        json.dump({"stamp": stamp(), "files": files}, open(os.path.join(d, prefix + "_STAMP.json"), "w"), indent=1)
    else:
        print(json.dumps(stamp(), indent=1))
