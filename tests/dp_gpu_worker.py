"""One rank of the data-parallel GPU test (tests/test_data_parallel_gpu.py starts two of these as fresh processes).

    python tests/dp_gpu_worker.py <rank> <world> <port> <backend> <case> <out_dir>

Every rank builds the same model, takes ITS rows of the same seeded global batches (train.shard_batch) and runs the
real `TrainStep` — HIP forward/backward on the device, the step's collectives through torch.distributed — for three
iterations.  It saves the reduced gradient of the first step, the per-step global losses and its parameters after the
last step; the parent compares them with a single-process TrainStep on the global batches.
backend "gloo": two processes share one GPU (RCCL refuses two ranks on one device), device buffers are staged through
host memory by GradientExchange.  backend "nccl" with world 1: the collectives are issued for real on a one-rank RCCL
communicator (always_collective), which exercises RCCL's stream hand-over around the HIP step on a single device."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]

CASES = {
    # name: (workload, model overrides, Shape overrides)
    "demo": ("demo", {}, dict(batch=7, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7,
                              max_target=10, ragged=True)),
    "demo_aux": ("demo", {"auxiliary_task": True},
                 dict(batch=7, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7, max_target=10,
                      ragged=True)),
    "compositional": ("compositional", {}, dict(batch=64, ragged=True)),
    "compositional_aux": ("compositional", {"auxiliary_task": True}, dict(batch=64, ragged=True)),
}
STEPS = 3


def build(case):
    import torch
    from multimodal_seq2seq_gscan_amd.config import model_kwargs
    from multimodal_seq2seq_gscan_amd.model import Model
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
    from weights import golden_weights
    workload, overrides, shape_kw = CASES[case]
    cfg = model_kwargs(workload, cnn_dropout_p=0.0, encoder_dropout_p=0.0, decoder_dropout_p=0.0, **overrides)
    model = Model(**cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in golden_weights(cfg, 31).items()}, strict=False)
    batches = [make_batch(Shape(**shape_kw), seed=900 + i) for i in range(STEPS)]
    return cfg, model.cuda(), batches


def run(model, batches, shard=None, **step_kw):
    """Three TrainStep iterations; returns (first reduced gradient, losses, final parameters, token counts)."""
    import torch
    from multimodal_seq2seq_gscan_amd.train import TrainStep, shard_batch
    grads = []
    step = TrainStep(model, learning_rate=1e-3, on_gradients=lambda g: grads.append(g.detach().cpu().clone()),
                     **step_kw)
    losses, tokens = [], []
    for batch in batches:
        if shard is not None:
            batch = shard_batch(batch, *shard)
        out = step({k: v.cuda() for k, v in batch.items()})
        losses.append(float(out["loss"].item()))
        tokens.append(float(out["tokens"].item()))
    torch.cuda.synchronize()
    return grads[0], losses, model.flat_parameters.detach().cpu().clone(), tokens, step


def main():
    rank, world, port, backend, case, out_dir = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4],
                                                 sys.argv[5], sys.argv[6])
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    cfg, model, batches = build(case)
    grad, losses, params, tokens, step = run(model, batches, shard=(rank, world), always_collective=(world == 1))
    assert step.exchange.collective and step.exchange.host_staged == (backend == "gloo")
    assert step.single_exchange == (not cfg["auxiliary_task"]) and not step.fused_loss
    native = step.exchange.comm is not None
    assert native == (backend == "nccl" and os.environ.get("GSCAN_NATIVE_ALLREDUCE", "1") != "0")
    torch.save({"grad": grad, "losses": losses, "params": params, "tokens": tokens, "native": native,
                "buckets": step.exchange.buckets if step._early_split else 1,
                "rows": [int(b["commands"].shape[0]) for b in batches]}, os.path.join(out_dir, f"rank{rank}.pt"))
    if native:
        step.exchange.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
