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
    """backend "nccl" IS RCCL on ROCm.  The step's float32 all-reduces go through the library's own communicator
    (`gscan_comm_init` from the process group, `gscan_allreduce_f32` on the step's stream: the worker asserts it was
    created and used); identity on one rank, but the launch sequence is the production one."""
    ranks = _launch(1, "nccl", case, tmp_path)
    assert ranks[0]["native"] is True
    _compare(ranks, _single_process(case))


@pytest.mark.parametrize("backend,world,case", [("gloo", 2, "compositional"), ("gloo", 2, "demo_aux"), ("nccl", 1, "demo"),
                                                ("nccl", 1, "compositional_aux")])
def test_two_bucket_gradient_exchange_matches_the_single_process_step(tmp_path, backend, world, case):
    """GSCAN_DP_BUCKETS=2: the early group of gradients (bridge, textual attention, decoder + the statistics) is all-reduced
    on a communication stream behind gscan_early_gradients_wait, the rest on the step's stream; the result must be what
    ONE all-reduce gives — two ranks sharing the device (gloo, staged through the host: the arithmetic and the order of the
    collectives) and one rank on the library's RCCL communicator (the production launch sequence and its stream hand-over)."""
    os.environ["GSCAN_DP_BUCKETS"] = "2"
    try:
        ranks = _launch(world, backend, case, tmp_path)
    finally:
        del os.environ["GSCAN_DP_BUCKETS"]
    assert all(r["buckets"] == 2 for r in ranks)
    _compare(ranks, _single_process(case))


def test_collectives_through_torch_distributed_rccl(tmp_path):
    """The fallback transport: torch.distributed's RCCL stream (GSCAN_NATIVE_ALLREDUCE=0)."""
    os.environ["GSCAN_NATIVE_ALLREDUCE"] = "0"
    try:
        ranks = _launch(1, "nccl", "demo", tmp_path)
    finally:
        del os.environ["GSCAN_NATIVE_ALLREDUCE"]
    assert ranks[0]["native"] is False
    _compare(ranks, _single_process("demo"))


def test_c_abi_communicator_without_torch_distributed():
    """gscan_comm_unique_id / gscan_comm_init / gscan_allreduce_f32 / gscan_comm_destroy by themselves (what a
    maintainer of the reference would bind): a one-rank communicator, a sum on the current stream and on a side
    stream, ordered with the kernels around it."""
    import ctypes as C
    from multimodal_seq2seq_gscan_amd import _lib
    lib = _lib.load()
    uid = C.create_string_buffer(_lib.COMM_ID_BYTES)
    _lib.check(lib.gscan_comm_unique_id(C.addressof(uid)), "unique_id")
    assert any(uid.raw)
    comm = C.c_void_p()
    _lib.check(lib.gscan_comm_init(C.byref(comm), 1, 0, C.addressof(uid)), "comm_init")
    x = torch.arange(440_279, dtype=torch.float32, device="cuda")
    want = (x * 2 + 1).cpu()
    x.mul_(2)
    _lib.check(lib.gscan_allreduce_f32(comm, x.data_ptr(), x.numel(), torch.cuda.current_stream().cuda_stream), "allreduce")
    x.add_(1)
    assert torch.equal(x.cpu(), want)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _lib.check(lib.gscan_allreduce_f32(comm, x.data_ptr(), x.numel(), side.cuda_stream), "allreduce")
    side.synchronize()
    assert torch.equal(x.cpu(), want)
    assert lib.gscan_allreduce_f32(None, x.data_ptr(), 4, None) != 0 and b"NULL" in lib.gscan_last_error()
    torch.cuda.synchronize()
    _lib.check(lib.gscan_comm_destroy(comm), "comm_destroy")


def test_dropout_streams_differ_by_rank_and_repeat_per_rank():
    """Philox streams keyed by (seed, rank, step).  Two ranks must not draw the same masks for their
    local rows; one rank must draw the same masks again from the same (seed, rank, step)."""
    from multimodal_seq2seq_gscan_amd.config import model_kwargs
    from multimodal_seq2seq_gscan_amd.model import Model

    def draws(rank, steps=2):
        torch.manual_seed(0)
        m = Model(**model_kwargs("compositional")).cuda().train()
        m.set_dropout_rank(rank)
        out = []
        for _ in range(steps):
            out.append([x.clone() for x in m._draw_masks(8, 10, 20, 36, torch.device("cuda"), materialize=True)])
        return out
    r0, r0_again, r1 = draws(0), draws(0), draws(1)
    for step in range(2):
        for a, b, c in zip(r0[step], r0_again[step], r1[step]):
            assert torch.equal(a, b)                               # reproducible per rank
            assert (a != c).float().mean().item() > 0.05           # p = 0.1 .. 0.3: independent draws differ often
    for a, b in zip(r0[0], r0[1]):
        assert not torch.equal(a, b)                               # and every step draws afresh
    with pytest.raises(ValueError):
        Model(**model_kwargs("demo")).set_dropout_rank(1 << 16)


def test_bench_self_launch_two_ranks_share_the_device():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the ranks before touching the device.
    Rehearsed here with --backend gloo (two ranks on the box's one GPU; RCCL needs a device per rank)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3",
                        "--warmup", "2", "--windows", "0", "--cpu-seconds", "0", "--batch", "32"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64 and out["value"] > 0
    assert out["config"]["gradient_exchange"] == "torch.distributed gloo"


def test_command_line_training_on_a_dataset_file_with_two_ranks(tmp_path):
    """`python -m seq2seq --mode=train` under two rank processes (what torch.distributed.run starts), on a generated
    dataset FILE: every rank reads the file, stages only its own rows of each global batch, the step's collectives
    run (gloo here: two ranks on the box's one device), evaluation is sharded over the ranks, rank 0 checkpoints."""
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, write_dataset_file
    data_dir = tmp_path / "data"
    data_dir.mkdir()
    write_dataset_file(str(data_dir / "dataset.txt"), {"train": 70, "dev": 12}, Shape(batch=1, max_command=8, max_target=10), seed=21)
    out = tmp_path / "out"
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GSCAN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
        procs.append(subprocess.Popen(
            [sys.executable, "-m", "seq2seq", "--mode", "train", "--data_directory", str(data_dir), "--output_directory", str(out),
             "--generate_vocabularies", "--training_batch_size", "16", "--max_training_iterations", "12", "--print_every", "4",
             "--evaluate_every", "6", "--max_decoding_steps", "10", "--embedding_dimension", "5", "--encoder_hidden_size", "20",
             "--decoder_hidden_size", "20", "--seed", "2"],
            env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=420)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r}:\n{logs[r][-3000:]}"
    assert "Iteration 00000012" in logs[0] and "Evaluation Accuracy" in logs[0]
    assert "Iteration" not in logs[1].replace("trained_iterations", "")          # only rank 0 logs progress
    # (a checkpoint is written only when the dev exact match improves, train.py:141-149: not after 12 iterations)
    assert os.path.exists(data_dir / "training_input_vocab.txt") and os.path.exists(data_dir / "training_target_vocab.txt")
    assert logs[0].count("Evaluation Accuracy") == 2 and "Finished training." in logs[0] and "Finished training." in logs[1]


def test_bench_under_a_launcher_environment_runs_the_nccl_path_on_one_rank():
    """The RCCL branch of bench.py on the box's one GPU: an explicit launcher environment (WORLD_SIZE=1, as
    torch.distributed.run would set it), --backend nccl and --always-collective, so that the process group is "nccl",
    the library's own communicator is created from it (collective transport decision included) and every step's
    all-reduce goes through gscan_allreduce_f32 on the step's stream.  The line must say so: rccl_nranks is what
    ncclCommCount reports for the communicator the step used."""
    import json
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl", "--always-collective",
                        "--steps", "5", "--warmup", "2", "--min-warmup-steps", "5", "--windows", "0", "--cpu-seconds", "0",
                        "--batch", "64"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["config"]["gradient_exchange"] == "gscan_allreduce_f32: RCCL on the step's stream"
    assert out["config"]["rccl_nranks"] == 1


def test_bench_strong_scaling_line_splits_a_fixed_global_batch():
    """--global-batch N (SURVEY.md 8d's secondary line): the ranks share a FIXED global batch and the line says
    "scaling": "strong".  Two ranks on the one device (gloo rehearsal)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--global-batch", "64",
                        "--steps", "3", "--warmup", "2", "--min-warmup-steps", "3", "--windows", "0", "--cpu-seconds", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 64 and out["n_gpus"] == 2
