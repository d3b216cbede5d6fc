"""`bench.py --gpus N` has to START N ranks (the flag was parsed and ignored).  Runs the launcher on CPU: two rank processes, gloo,
no model (`--launch-only`); one JSON line comes out, from rank 0, and says two ranks met."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=240):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def test_gpus_flag_starts_that_many_ranks():
    r = _run(["--gpus", "2", "--launch-only"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                         # exactly ONE line on stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["allreduce_ok"] is True and j["backend"] == "gloo"


def test_a_failing_rank_fails_the_launch():
    # rank 1 cannot parse its arguments (an unknown flag injected through the environment is not possible, so: a world size the ranks refuse)
    r = _run(["--gpus", "2", "--launch-only", "--global-batch", "3", "--no-such-flag"])
    assert r.returncode != 0
    assert not r.stdout.decode().strip()


def test_under_a_launcher_the_flag_does_not_fork_again():
    # WORLD_SIZE in the environment = a launcher (torch.distributed.run) already started the ranks: this process IS rank 0 of 1
    r = _run(["--gpus", "1", "--launch-only"], env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577"})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    j = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 1


def test_the_launcher_process_never_imports_torch():
    code = ("import sys; sys.argv = ['bench.py', '--gpus', '2', '--launch-only']; sys.path.insert(0, %r); import bench\n"
            "try:\n    bench.main()\nexcept SystemExit as e:\n    assert e.code == 0, e.code\n"
            "assert 'torch' not in sys.modules, 'the launcher loaded torch (and with it the HIP runtime) before starting the ranks'\n") % ROOT
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, "-c", code], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
