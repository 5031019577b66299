"""Data-parallel host logic on CPU with gloo, world_size 2 (the GPU path uses the same class over RCCL).

Each rank owns half of a global batch, computes its local [sum NLL, tokens, sum aux NLL, rows] and its local
gradient seeded with the values `GradientExchange.seeds` returns; after the flat all-reduce every rank must
hold the gradient of the reference's single-process loss on the GLOBAL batch, including when the shards have
different numbers of non-pad tokens (a plain average of per-rank means would be wrong there)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, auxiliary: bool, out_dir: str):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from multimodal_seq2seq_gscan_amd.config import model_kwargs
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
    from multimodal_seq2seq_gscan_amd.train import GradientExchange, shard_batch
    from oracle import seq2seq_oracle as oracle            # the checker: stands in for the HIP step on CPU
    from weights import golden_weights

    cfg = model_kwargs("demo", auxiliary_task=auxiliary)
    params = {k: torch.from_numpy(v) for k, v in golden_weights(cfg, 3).items()}
    shape = Shape(batch=6, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7, max_target=10,
                  ragged=True)
    full = make_batch(shape, seed=77)
    mine = shard_batch(full, rank, world)
    w = 0.3

    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    logp, aux = oracle.forward(leaves, mine["commands"], mine["cmd_lengths"], mine["world"], mine["targets"],
                               conditional=True, auxiliary=auxiliary)
    nll_sum, tokens = oracle.sequence_loss(logp, mine["targets"], reduction="sum")
    stats = torch.zeros(4)
    stats[0], stats[1], stats[3] = nll_sum.item(), float(tokens), float(mine["commands"].shape[0])
    aux_sum = None
    if auxiliary:
        aux_sum = -aux.gather(1, mine["target_positions"].view(-1, 1)).sum()
        stats[2] = aux_sum.item()
    exchange = GradientExchange()
    assert exchange.world_size == world and exchange.rank == rank
    stats, seq_seed, aux_seed, loss = exchange.seeds(stats, w, auxiliary)
    local = nll_sum * seq_seed
    if auxiliary:
        local = local + aux_sum * aux_seed
    local.backward()
    names = list(params)
    flat = torch.cat([leaves[k].grad.reshape(-1) if leaves[k].grad is not None else torch.zeros(params[k].numel())
                      for k in names])
    exchange.all_reduce(flat)

    ref_loss, ref_grads, _ = oracle.loss_and_grads(params, full, conditional=True, auxiliary=auxiliary,
                                                   weight_target_loss=w)
    ref_flat = torch.cat([ref_grads[k].reshape(-1) for k in names])
    ok = (abs(loss.item() - ref_loss.item()) < 1e-5 and torch.allclose(flat, ref_flat, atol=2e-6, rtol=1e-4)
          and stats[3].item() == 6.0)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(f"{int(ok)} loss={loss.item():.6f} ref={ref_loss.item():.6f} "
                f"maxerr={(flat - ref_flat).abs().max().item():.2e} tokens={stats[1].item()}")
    dist.destroy_process_group()


@pytest.mark.parametrize("auxiliary", [False, True])
def test_two_ranks_reproduce_the_global_batch_gradient(tmp_path, auxiliary):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, auxiliary, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        text = open(tmp_path / f"rank{r}.txt").read()
        assert text.startswith("1 "), text


def test_shard_batch_covers_every_row_once():
    from multimodal_seq2seq_gscan_amd.synthetic import S0_DEMO, make_batch
    from multimodal_seq2seq_gscan_amd.train import shard_batch
    batch = make_batch(S0_DEMO._replace(batch=7) if hasattr(S0_DEMO, "_replace") else S0_DEMO)
    B = batch["commands"].shape[0]
    for world in (1, 2, 3):
        parts = [shard_batch(batch, r, world) for r in range(world)]
        assert sum(p["commands"].shape[0] for p in parts) == B
        assert torch.equal(torch.cat([p["targets"] for p in parts]), batch["targets"])


def _worker_single_exchange(rank: int, world: int, port: int, out_dir: str):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from multimodal_seq2seq_gscan_amd.config import model_kwargs
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
    from multimodal_seq2seq_gscan_amd.train import GradientExchange, shard_batch
    from oracle import seq2seq_oracle as oracle            # the checker: stands in for the HIP step on CPU
    from weights import golden_weights

    cfg = model_kwargs("demo", auxiliary_task=False)
    params = {k: torch.from_numpy(v) for k, v in golden_weights(cfg, 3).items()}
    shape = Shape(batch=6, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7, max_target=10,
                  ragged=True)
    full = make_batch(shape, seed=78)
    mine = shard_batch(full, rank, world)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    logp, _ = oracle.forward(leaves, mine["commands"], mine["cmd_lengths"], mine["world"], mine["targets"],
                             conditional=True, auxiliary=False)
    nll_sum, tokens = oracle.sequence_loss(logp, mine["targets"], reduction="sum")
    nll_sum.backward()                                      # what gscan_backward_nll(sum_reduction=1) produces
    names = list(params)
    store = torch.cat([leaves[k].grad.reshape(-1) if leaves[k].grad is not None else torch.zeros(params[k].numel())
                       for k in names] + [torch.tensor([nll_sum.item(), float(tokens), 0.0,
                                                        float(mine["commands"].shape[0])])])
    store, count, loss = GradientExchange().mean_from_sums(store)
    flat = store[:-4] / count                               # what gscan_adam_step_mean divides by
    ref_loss, ref_grads, _ = oracle.loss_and_grads(params, full, conditional=True, auxiliary=False)
    ref_flat = torch.cat([ref_grads[k].reshape(-1) for k in names])
    ok = (abs(loss.item() - ref_loss.item()) < 1e-5 and torch.allclose(flat, ref_flat, atol=2e-6, rtol=1e-4)
          and store[-1].item() == 6.0)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(f"{int(ok)} loss={loss.item():.6f} ref={ref_loss.item():.6f} "
                f"maxerr={(flat - ref_flat).abs().max().item():.2e} tokens={count.item()}")
    dist.destroy_process_group()


def test_two_ranks_single_exchange_of_sum_gradients_and_statistics(tmp_path):
    """The one-collective step (no auxiliary loss): sum-loss gradients and the statistics travel together."""
    world, port = 2, _free_port()
    mp.spawn(_worker_single_exchange, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        text = open(tmp_path / f"rank{r}.txt").read()
        assert text.startswith("1 "), text
