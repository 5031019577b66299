"""`python bench.py --gpus N` without a launcher starts its own rank processes (the driver's multi-GPU tier may call
it either way).  CPU-only: the rank plumbing is exercised with --backend gloo --dry-launch; the device never enters."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*flags, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *flags], env=e, capture_output=True, text=True, timeout=300)


def test_self_launch_starts_n_ranks_and_relays_one_json_line():
    r = _run("--gpus", "2", "--backend", "gloo", "--dry-launch")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {"dry_launch": True, "n_gpus": 2, "backend": "gloo", "rank_sum": 3.0}


def test_more_ranks_than_devices_is_one_clear_error():
    """No GPU in this container: --gpus 2 over RCCL must refuse before starting anything."""
    import torch
    if torch.cuda.device_count() >= 2:
        return
    r = _run("--gpus", "2")
    assert r.returncode == 2 and r.stdout.strip() == ""
    assert "HIP device(s) are visible" in r.stderr and r.stderr.count("\n") <= 2


def test_a_failing_rank_fails_the_launch():
    """Ranks that cannot run (no device here, no CPU fallback in the product path) make the parent exit non-zero."""
    import torch
    if torch.cuda.is_available():
        return
    r = _run("--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and "rank exit codes" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_launched_by_torch_distributed_run_the_ranks_do_not_relaunch():
    """Under a launcher (WORLD_SIZE set) bench.py is a plain rank process; a mismatch with --gpus is an error."""
    r = _run("--gpus", "2", "--dry-launch", "--backend", "gloo", env={"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr
