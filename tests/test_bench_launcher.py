"""bench.py --gpus N must run N ranks or fail: the launcher starts torch.distributed.run as a child process (the
parent never touches a GPU), rank 0 reports the size of the process group it actually joined, and a rank whose
WORLD_SIZE differs from --gpus exits non-zero.  Runs on CPU: --dry-run joins the group over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra=None):
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")  # gloo even on a GPU box
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert lines, stdout
    return json.loads(lines[-1])


def test_gpus_2_launches_two_ranks():
    r = run(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = last_json(r.stdout)
    assert out["n_gpus"] == 2 and out["ranks_counted"] == 2 and out["backend"] == "gloo"
    # the preflight carries the real run's two messages (packed fp64 ELBO buffer, one 128 MiB gradient bucket) and says who
    # took part
    assert out["ok"] and out["bucket_ranks_counted"] == 2 and set(out["allreduce_ms"]) == {"packed_fp64_66", "bucket_128MiB"}
    assert [r["rank"] for r in out["ranks"]] == [0, 1]


def test_single_rank_needs_no_launcher():
    r = run(["--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert last_json(r.stdout)["n_gpus"] == 1


def test_world_size_mismatch_is_an_error():
    # a 1-rank environment under an N-rank flag must not print a line at all
    r = run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_strong_scaling_of_ten_samples_over_eight_ranks_through_the_launcher():
    """BASELINE configs[4]'s 8-GPU leg as the driver will launch it (`--strong --samples 10 --gpus 8`): eight rank processes
    come up, the preflight counts all of them over both message kinds, and the ranks report the uneven shards
    2, 2, 1, 1, 1, 1, 1, 1 of the step's ten global sample indices."""
    r = run(["--gpus", "8", "--strong", "--samples", "10", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = last_json(r.stdout)
    assert out["ok"] and out["n_gpus"] == 8 and out["ranks_counted"] == 8 and out["bucket_ranks_counted"] == 8
    assert out["scaling"] == "strong" and out["samples_per_step"] == 10
    ranks = out["ranks"]
    assert [q["rank"] for q in ranks] == list(range(8))
    assert [q["samples"] for q in ranks] == [2, 2, 1, 1, 1, 1, 1, 1]
    assert [q["first_sample"] for q in ranks] == [0, 2, 4, 5, 6, 7, 8, 9]
    # the IPC setting multi-process GPU work depends on is logged and echoed by every rank, with where it came from
    assert all(q["ipc_mode_legacy"] is not None for q in ranks)


def test_ipc_mode_is_defaulted_logged_and_overridable():
    r = run(["--gpus", "2", "--dry-run"], {"HSA_ENABLE_IPC_MODE_LEGACY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert all(q["ipc_mode_legacy"] == "1 (environment)" for q in last_json(r.stdout)["ranks"])
    assert "HSA_ENABLE_IPC_MODE_LEGACY=1 (from the environment)" in r.stderr
    env = dict(os.environ)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                        env={k: v for k, v in dict(env, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="").items()
                             if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")},
                        capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert all(q["ipc_mode_legacy"] == "0 (bench.py default)" for q in last_json(r2.stdout)["ranks"])
    assert "defaulting to 0" in r2.stderr
