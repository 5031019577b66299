"""The data-parallel training step on the GPU (the reference has no distributed code, train.py:24,65).

Two fresh processes share the one device of the test box (gloo; device buffers staged through the host by
GradientExchange) and run the REAL `TrainStep` — `gscan_forward`, `gscan_backward_nll(sum_reduction=1)` +
`gscan_adam_step_mean` in the one-collective form, the statistics all-reduce + seeded backward + gradient all-reduce
with the auxiliary loss — on unequal shards (7 rows -> 3 + 4; ragged targets, so token counts differ).  Their global
loss, reduced gradient and post-Adam parameters must equal a single-process TrainStep on the global batch.  A third
case issues the same collectives on a one-rank RCCL communicator (backend "nccl"), the production transport.
Run: pytest -m gpu."""
import os
import socket
import subprocess
import sys

import pytest
import torch

import dp_gpu_worker as worker

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, backend, case, out_dir):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_gpu_worker.py"), str(r), str(world),
                               str(port), backend, case, str(out_dir)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    return [torch.load(os.path.join(out_dir, f"rank{r}.pt"), weights_only=False) for r in range(world)]


def _single_process(case):
    cfg, model, batches = worker.build(case)
    grad, losses, params, tokens, step = worker.run(model, batches)
    assert step.fused_loss and not step.exchange.collective
    return grad, losses, params, tokens


def _compare(ranks, ref):
    grad, losses, params, tokens = ref
    for r, got in enumerate(ranks):
        for i, (a, b) in enumerate(zip(got["losses"], losses)):
            assert abs(a - b) < 1e-5, (r, i, a, b)
        assert got["tokens"] == tokens, (got["tokens"], tokens)          # GLOBAL live-token counts
        err = (got["grad"] - grad).abs().max().item()
        assert torch.allclose(got["grad"], grad, atol=1e-6, rtol=1e-4), f"rank {r}: reduced gradient off by {err:.3e}"
        # Adam turns ~1e-9 rounding differences of tiny gradients into steps of the size of the learning rate, so
        # the parameters are compared where the first gradient is solid, and bounded everywhere
        solid = grad.abs() > 1e-5
        assert (got["params"] - params).abs()[solid].max().item() < 2e-5
        assert (got["params"] - params).abs().max().item() < 3.5e-3
    for got in ranks[1:]:
        assert torch.equal(got["params"], ranks[0]["params"])             # replicas stay bit-identical
        assert torch.equal(got["grad"], ranks[0]["grad"])


@pytest.mark.parametrize("case", ["demo", "demo_aux", "compositional", "compositional_aux"])
def test_two_ranks_on_one_device_match_the_single_process_step(tmp_path, case):
    ranks = _launch(2, "gloo", case, tmp_path)
    if case.startswith("demo"):
        assert ranks[0]["rows"] == [7, 7, 7]
    _compare(ranks, _single_process(case))


def test_four_ranks_on_one_device(tmp_path):
    """Four ranks (7 rows -> shards of 1, 2, 2, 2), the two-collective auxiliary form: the same global gradient."""
    ranks = _launch(4, "gloo", "demo_aux", tmp_path)
    _compare(ranks, _single_process("demo_aux"))


@pytest.mark.parametrize("case", ["demo", "demo_aux"])
def test_collectives_on_a_one_rank_rccl_communicator(tmp_path, case):
    """backend "nccl" IS RCCL on ROCm: the step's all-reduces are launched by RCCL on its own stream between the
    library's launches (identity on one rank, but the stream hand-over is the production one)."""
    ranks = _launch(1, "nccl", case, tmp_path)
    _compare(ranks, _single_process(case))
