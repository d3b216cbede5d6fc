"""Evidence stamps: a profile file is quoted by bench.py only if it was measured on the sources the library is built from."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import stamp  # noqa: E402


def _copy_tree(dst):
    for rel in ("kasportsformer_amd/csrc", "include"):
        os.makedirs(os.path.join(dst, rel), exist_ok=True)
    for f in stamp.source_files(ROOT):
        shutil.copy(f, os.path.join(dst, os.path.relpath(f, ROOT)))


def test_editing_a_kernel_without_refreshing_makes_every_stamped_file_stale(tmp_path):
    tree = str(tmp_path / "tree")
    _copy_tree(tree)
    st = stamp.stamp(tree)
    assert st["source_sha256"] == stamp.source_sha256(ROOT)          # same sources, same identity: the stamp does not depend on where the tree lives
    prof = tmp_path / "profiles"
    prof.mkdir()
    j = prof / "r5_pmc_traffic.json"
    j.write_text(json.dumps({"kernels": {}}))
    stamp.embed(str(j), st)
    jl = prof / "r5_configs.jsonl"
    jl.write_text('{"config": "x", "clips_per_s": 1}\n')
    stamp.embed(str(jl), st)
    csv = prof / "r5_train_kernel_stats.csv"
    csv.write_text("Name,Calls\n")
    other = prof / "r5_unlisted.csv"
    (prof / "r5_STAMP.json").write_text(json.dumps({"stamp": st, "files": ["r5_train_kernel_stats.csv"]}))
    other.write_text("x\n")
    old = prof / "r4_pmc_step.json"
    old.write_text(json.dumps({"hbm_GB_per_step": 212.7}))            # a pre-round-5 file: no stamp
    assert stamp.is_fresh(str(j), tree) and stamp.is_fresh(str(jl), tree) and stamp.is_fresh(str(csv), tree)
    assert not stamp.is_fresh(str(other), tree)                       # not covered by the sidecar
    assert not stamp.is_fresh(str(old), tree)
    assert json.loads(jl.read_text().splitlines()[0])["clips_per_s"] == 1          # the data lines are untouched
    with open(os.path.join(tree, "kasportsformer_amd", "csrc", "k_mlp3.hip"), "a") as f:
        f.write("// one more comment\n")                              # edit a kernel, do not refresh
    assert not stamp.is_fresh(str(j), tree) and not stamp.is_fresh(str(jl), tree) and not stamp.is_fresh(str(csv), tree)


def test_bench_reports_stale_files_and_omits_their_figures(tmp_path, monkeypatch):
    import bench
    monkeypatch.setattr(bench, "STALE", [])
    f = tmp_path / "r5_pmc_traffic.json"
    f.write_text(json.dumps({"kernels": {"k_mlp_bwd_s": {"hbm_bytes": 1}, "k_lnbwd_sum4_fin": {"hbm_bytes": 2}},
                             "stamp": {"source_sha256": "0" * 64}}))
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(f))
    M = bench.BATCH_PER_GPU * bench.T * 17
    assert bench.pmc_traffic("k_mlp_bwd_s(+lnbwd_sum4_fin)", M) is None          # stale: omitted ...
    assert bench.STALE == ["r5_pmc_traffic.json"]                                 # ... and named
    j = json.loads(f.read_text())
    j["stamp"] = stamp.stamp(ROOT)
    f.write_text(json.dumps(j))
    assert bench.pmc_traffic("k_mlp_bwd_s(+lnbwd_sum4_fin)", M) == 3
