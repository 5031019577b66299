"""`python bench.py --gpus N` without a launcher starts its own rank processes (the driver's multi-GPU tier may call
it either way).  CPU-only: the rank plumbing is exercised with --backend gloo --dry-launch; the device never enters."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*flags, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT",
                                                           "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES",
                                                           "CUDA_VISIBLE_DEVICES")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *flags], env=e, capture_output=True, text=True, timeout=300)


def test_self_launch_starts_n_ranks_and_relays_one_json_line():
    r = _run("--gpus", "2", "--backend", "gloo", "--dry-launch")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {"dry_launch": True, "n_gpus": 2, "backend": "gloo", "rank_sum": 3.0}


def _fake_topology(tmp_path, gpus):
    """A directory laid out like /sys/class/kfd/kfd/topology/nodes: node 0 is the CPU, then `gpus` GPU nodes."""
    for i in range(gpus + 1):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {64 if i == 0 else 0}\nsimd_count {0 if i == 0 else 1024}\n")
    return str(tmp_path)


def test_more_ranks_than_devices_is_one_clear_error(tmp_path):
    """The nccl branch of the launcher: the device count comes from the KFD topology (no HIP call in the parent), and
    --gpus 2 on a one-GPU machine is refused before anything starts."""
    r = _run("--gpus", "2", env={"GSCAN_KFD_TOPOLOGY": _fake_topology(tmp_path, 1)})
    assert r.returncode == 2 and r.stdout.strip() == ""
    assert "only 1 GPU(s) are visible" in r.stderr and r.stderr.count("\n") <= 2


def test_visible_devices_lists_narrow_the_count(tmp_path):
    topo = _fake_topology(tmp_path, 4)
    r = _run("--gpus", "2", env={"GSCAN_KFD_TOPOLOGY": topo, "HIP_VISIBLE_DEVICES": "2"})
    assert r.returncode == 2 and "only 1 GPU(s) are visible" in r.stderr


def test_nccl_parent_branch_starts_the_ranks_without_touching_the_device(tmp_path):
    """Enough GPUs by the topology: the parent goes on to start the ranks.  Here they have no device and exit at
    once; the parent reports the first failure and never prints a JSON line.  GSCAN_FORBID_HIP_IN_PARENT makes any
    torch.cuda call in the PARENT an error, so this run proves the nccl branch makes none."""
    import torch
    if torch.cuda.is_available():
        return
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0",
             env={"GSCAN_KFD_TOPOLOGY": _fake_topology(tmp_path, 8), "GSCAN_FORBID_HIP_IN_PARENT": "1"})
    assert r.returncode != 0 and "rank exit codes" in r.stderr, r.stderr[-2000:]
    assert "torch.cuda touched in the launcher parent" not in r.stderr
    assert "needs the HIP device" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


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
