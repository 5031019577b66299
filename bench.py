"""Benchmark of the training hot path: examples/s of one full training step
(forward + loss + backward + gradient all-reduce + Adam/LR step) on synthetic compositional_splits
batches, one process per GPU.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus N ...        (no launcher: starts N fresh rank processes itself, before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task description): `value` is the whole-job
examples/s with inputs resident in HBM, weak scaling (256 rows per GPU).  `roofline` describes the
dominant kernel family, timed with HIP events on its launch stream inside a second pass over the
same K steps; `cpu_baseline` is the CPU oracle (a port of the reference's CPU path) timed on the
host cores of the same box on the same workload.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# fp32 matrix peak of MI355X (v_mfma_f32_*_f32 = the fp32 vector rate), MI355X_MICROARCH.md "Chip-level parameters"
PEAK_FP32_TFLOPS = 157.3
PROBES = ("decoder_forward", "decoder_backward", "encoder_forward", "encoder_backward", "gemm", "conv_forward",
          "conv_backward", "keys_backward")
# HBM-side bytes per launch of the GEMM family cannot be read from inside the process (FETCH_SIZE / WRITE_SIZE are
# rocprofv3 PMC counters, collected in two separate passes over this same command by tools/gpu_round.sh, which writes
# this file together with a hash of the kernel sources it profiled).  The number is reported only when that hash
# equals the hash of the sources of the library that is running; a stale profile yields null and a warning.
PMC_TRAFFIC = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")


def source_hash() -> str:
    """sha256 over the kernel sources and the public header (what libgscan_hip.so is built from)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "multimodal_seq2seq_gscan_amd", "csrc")
    files = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    for path in [os.path.join(csrc, f) for f in files] + [os.path.join(ROOT, "include", "gscan_hip.h")]:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_family: str, workload: str):
    """(bytes per launch of the GEMM family or None, provenance string, {kernel: bytes per launch} of every profiled kernel)."""
    if kernel_family != "gemm" or not os.path.exists(PMC_TRAFFIC):
        return None, "no PMC profile for this kernel family", {}
    try:
        with open(PMC_TRAFFIC) as f:
            prof = json.load(f)
        if prof.get("source_sha") != source_hash():
            print(f"bench.py: WARNING: {PMC_TRAFFIC} was collected on other kernel sources "
                  f"({prof.get('source_sha')} != {source_hash()}): roofline.traffic is null; rerun tools/gpu_round.sh",
                  file=sys.stderr, flush=True)
            return None, "stale PMC profile (kernel sources changed since it was collected)", {}
        if prof.get("workload", "compositional") != workload:
            return None, f"PMC profile is of workload {prof.get('workload')}", {}
        return round(float(prof["traffic_bytes_per_launch"])), f"profiles/pmc_traffic_latest.json ({prof.get('tag', '?')})", \
            {k: round(float(v["traffic_bytes_per_launch"])) for k, v in prof.get("kernels", {}).items()}
    except (OSError, ValueError, KeyError) as e:
        return None, f"unreadable PMC profile: {e}", {}


def step_traffic(workload: str, mflop_per_example: float, B: int):
    """Measured HBM-side bytes of ONE step over every profiled kernel (PMC passes of tools/gpu_round.sh, bound to the
    kernel sources by hash) next to SURVEY.md 8d's algorithmic figure (0.1-0.15 MB per example at T = 20)."""
    _, source, per = pmc_traffic("gemm", workload)
    if not per:
        return None
    # launches per step come from the kernel trace of the SAME PMC pass (tools/pmc_traffic.py: launches of the kernel /
    # launches of the optimiser), not from an assumed schedule; a profile without them yields null, not a guess
    try:
        with open(PMC_TRAFFIC) as f:
            kernels = json.load(f).get("kernels", {})
        launches = {k: kernels[k]["launches_per_step"] for k in per}
    except (OSError, ValueError, KeyError):
        return None
    if any(v is None for v in launches.values()):
        return None
    total = sum(v * launches[k] for k, v in per.items())
    return {"measured_bytes_per_step": round(total), "kernels_profiled": sorted(per),
            "launches_per_step": launches,
            "algorithmic_bytes_per_step": [round(0.10e6 * B), round(0.15e6 * B)],
            "measured_over_algorithmic": round(total / (0.125e6 * B), 2), "source": source}


def algorithmic_mflop_per_example(cfg: dict, G: int, L: int, T: int) -> float:
    """F_train of SURVEY.md §8(d): 6 x forward MACs, valid convolution taps only."""
    C_, Co, k3 = cfg["num_cnn_channels"], cfg["cnn_hidden_num_channels"], cfg["cnn_kernel_size"]
    E, He, H, V = cfg["embedding_dimension"], cfg["encoder_hidden_size"], cfg["decoder_hidden_size"], \
        cfg["target_vocabulary_size"]
    valid = lambda k: sum(1 for i in range(G) for j in range(G) if abs(i - j) <= k // 2)
    mac = C_ * Co * sum(valid(k) ** 2 for k in (1, 5, k3))
    mac += G * G * 3 * Co * H + 2 * L * 4 * He * (E + He) + L * He * H + He * H
    step = H * H + 2 * L * H + (2 * H * H if cfg["conditional_attention"] else 0) + H * H + 2 * G * G * H \
        + 4 * H * 4 * H + 4 * H * H + H * V
    return 6.0 * (mac + T * step) / 1e6


CPU_ANCHOR = os.path.join(ROOT, "profiles", "cpu_ref_vs_oracle_latest.json")


def reference_over_port():
    """{workload: reference examples/s / oracle examples/s, ...} as last measured in the build container, with its date,
    thread count and the oracle source's hash; None when the committed anchor is missing or of another oracle."""
    try:
        with open(CPU_ANCHOR) as f:
            anchor = json.load(f)
    except (OSError, ValueError):
        return None
    with open(os.path.join(ROOT, "oracle", "seq2seq_oracle.py"), "rb") as f:
        current = hashlib.sha256(f.read()).hexdigest()[:16]
    anchor["oracle_unchanged_since"] = anchor.get("oracle_sha") == current
    return anchor


def cpu_baseline(cfg: dict, shape, budget_s: float) -> dict:
    """The CPU oracle (oracle/seq2seq_oracle.py, a restatement of the reference's CPU path) on this box's cores:
    the same step definition (forward, loss, backward, Adam + LR) on the same synthetic workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from multimodal_seq2seq_gscan_amd.synthetic import make_batch
    from oracle import seq2seq_oracle as oracle
    from weights import golden_weights
    # The path issues ~20k small ATen ops per step: beyond a few dozen threads it gets slower, not faster
    # (256 threads: 0.9 examples/s on the MI355X host).  The thread count is swept below and the best one reported.
    allowed = len(os.sched_getaffinity(0))
    params = {k: torch.from_numpy(v) for k, v in golden_weights(cfg, 1).items()}
    names = list(params)
    m = [torch.zeros_like(params[k]) for k in names]
    v = [torch.zeros_like(params[k]) for k in names]
    batch = make_batch(shape, 1234)
    p = (cfg["cnn_dropout_p"], cfg["encoder_dropout_p"], cfg["decoder_dropout_p"])
    B, L = batch["commands"].shape
    T = batch["targets"].shape[1]
    G = batch["world"].shape[1]

    def one(step):
        masks = (torch.nn.functional.dropout(torch.ones(B, G * G, 3 * cfg["cnn_hidden_num_channels"]), p[0]),
                 torch.nn.functional.dropout(torch.ones(B, L, cfg["embedding_dimension"]), p[1]),
                 torch.nn.functional.dropout(torch.ones(B, T, cfg["decoder_hidden_size"]), p[2]))
        _, g, _ = oracle.loss_and_grads(params, batch, conditional=cfg["conditional_attention"],
                                        auxiliary=cfg["auxiliary_task"], masks=masks)
        oracle.adam_step([params[k] for k in names], [g[k] for k in names], m, v, step, 1e-3)

    sweep, step_no = {}, 1
    for cores in sorted({c for c in (8, 16, 32) if c <= allowed} or {allowed}):
        torch.set_num_threads(cores)
        one(step_no)                         # warm-up at this thread count (thread pools, allocator)
        t0 = time.perf_counter()
        one(step_no + 1)
        one(step_no + 2)
        sweep[cores] = 2 * B / (time.perf_counter() - t0)
        step_no += 3
    cores = max(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    one(step_no)
    n, t0 = 0, time.perf_counter()
    while True:
        one(step_no + 1 + n)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 64:
            break
    return {"value": round(n * B / el, 1), "unit": "examples/s", "cores": cores, "kind": "port",
            # how far the port is from the reference it restates: the IMPORTED reference timed beside the oracle in the
            # build container (tools/cpu_ref_vs_oracle.py --json; the reference cannot travel to the GPU box)
            "reference_over_port": reference_over_port(),
            "thread_sweep": {str(c): round(v, 1) for c, v in sweep.items()},
            "sample": f"{n} steps of the same workload (B={B}, T={T}) at the best thread count of the sweep, after "
                      f"warm-up, torch {torch.__version__} CPU, {allowed} cores allowed"}


def batcher_leg(args, cfg: dict, n_params: int) -> dict:
    """End-to-end examples/s with the dataset reader and batcher in the loop: a dataset FILE of
    --batcher-examples examples in the reference's format is generated, read by GroundedScanDataset, shuffled, and
    every batch is gathered on the host, copied to the device (uint8 world, one pinned slab) and trained on.  The
    same model dims / step as the headline number; compared with a run of the same batches resident in HBM."""
    import tempfile
    from multimodal_seq2seq_gscan_amd.dataset import BatchStager, GroundedScanDataset
    from multimodal_seq2seq_gscan_amd.model import Model
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, write_dataset_file
    from multimodal_seq2seq_gscan_amd.train import TrainStep
    B = args.batch
    keys = ("commands", "cmd_lengths", "world", "targets", "tgt_lengths", "target_positions")
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "dataset.txt")
        t0 = time.perf_counter()
        write_dataset_file(path, {"train": args.batcher_examples}, Shape(batch=1, max_command=args.command_length,
                                                                         max_target=args.target_length), seed=7)
        t_write = time.perf_counter() - t0
        t0 = time.perf_counter()
        data = GroundedScanDataset(path, tmp, k=0, split="train", generate_vocabulary=True)
        data.read_dataset()
        t_read = time.perf_counter() - t0
    cfg = dict(cfg, input_vocabulary_size=data.input_vocabulary_size, target_vocabulary_size=data.target_vocabulary_size,
               num_cnn_channels=data.image_channels)
    torch.manual_seed(42)
    model = Model(**cfg).cuda()
    step = TrainStep(model, learning_rate=1e-3)
    stager = BatchStager(torch.device("cuda"), data.slab_bytes(B))
    results = {}
    for bucket in (0, 8):
        # warm-up over the shapes this order produces (allocator blocks, cached slab views, kernel attributes), then
        # a fresh shuffle for the timed run
        data.shuffle_data(bucket_batches=bucket, batch_size=B)
        keep = []                                # clones of 64 full batches of this order (taken OUTSIDE the timed loop)
        for i, b in enumerate(data.batches(B, stager=stager)):
            if b["commands"].shape[0] == B:
                step({k: b[k] for k in keys})
                if len(keep) < 64 and i % 4 == 0:
                    keep.append({k: b[k].clone() for k in keys})
            if i >= 300:      # (five batches for the reference order, as until round 4, left first-touch costs in the timed loop)
                break
        data.shuffle_data(bucket_batches=bucket, batch_size=B)
        it = data.batches(B, stager=stager)
        torch.cuda.synchronize()
        # `resident`: the SAME kind of batches kept in HBM — clones of 64 full batches of this order (from the warm-up pass), cycled
        # through for as many steps (with length buckets every batch has its own padded lengths: repeating ONE batch, as
        # until round 5, compared the file-fed run with whatever length its last batch happened to have)
        n, rows, t0 = 0, 0, time.perf_counter()
        for b in it:
            if b["commands"].shape[0] != B:            # the split's short trailing batch
                continue
            step({k: b[k] for k in keys})
            n, rows = n + 1, rows + B
            if n >= max(4 * args.steps, 300):      # at least 300 batches (0.15 s): a 40 ms loop is at the mercy of one hiccup
                break
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        for i in range(max(args.warmup, len(keep))):
            step(keep[i % len(keep)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            step(keep[i % len(keep)])
        torch.cuda.synchronize()
        el_res = time.perf_counter() - t0
        results["length_buckets_8" if bucket else "reference_order"] = {
            "examples_per_s": round(rows / el, 1), "ms_per_step": round(1e3 * el / n, 4), "steps": n,
            "resident_examples_per_s": round(rows / el_res, 1), "fraction_of_resident": round(el_res / el, 3),
            "mean_target_length": round(sum(int(b["targets"].shape[1]) for b in keep) / len(keep), 2)}
    return {"examples": data.num_examples, "file_write_s": round(t_write, 1), "file_read_s": round(t_read, 1),
            "h2d_bytes_per_batch": data.slab_bytes(B), "parameters": model.parameter_count, **results,
            "note": "ragged lengths from the file (every batch padded to ITS longest rows, gSCAN_dataset.py:200-220); "
                    "resident = clones of the run's first 64 full batches cycled from HBM; GSCAN_BATCHER_THREAD=1 moves "
                    "the host gather to a worker thread a batch ahead (measured slower: interpreter-lock contention)"}


def visible_gpu_count(sysfs: str = ""):
    """GPUs this process tree would see, WITHOUT a HIP / HSA call (torch.cuda.device_count() falls through to
    hipGetDeviceCount on this build, which initialises the runtime in the launcher parent): the KFD topology lists
    one node per agent, GPUs are the nodes with simd_count > 0; the *_VISIBLE_DEVICES lists, where set, narrow that.
    None when the topology cannot be read — then the children report a bad LOCAL_RANK themselves."""
    # GSCAN_KFD_TOPOLOGY: another directory of the same layout (tests)
    sysfs = sysfs or os.environ.get("GSCAN_KFD_TOPOLOGY", "/sys/class/kfd/kfd/topology/nodes")
    try:
        nodes = sorted(os.listdir(sysfs), key=lambda n: int(n) if n.isdigit() else 1 << 30)
    except OSError:
        return None
    gpus = 0
    for node in nodes:
        try:
            with open(os.path.join(sysfs, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            gpus += 1
    if gpus == 0:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        listed = os.environ.get(var)
        if listed is not None:
            gpus = min(gpus, len([x for x in listed.split(",") if x.strip() != ""]))
    return gpus


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh processes of this same command with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relay rank 0's JSON line, fail as soon as any rank fails.  The
    parent makes NO HIP call (the device count comes from sysfs); the children are ordinary rank processes, exactly
    what torch.distributed.run would have started."""
    import socket
    import subprocess
    import threading
    n = args.gpus
    if os.environ.get("GSCAN_FORBID_HIP_IN_PARENT") == "1":     # tests: any torch.cuda call below this line raises
        def _forbidden(*_a, **_k):
            raise RuntimeError("torch.cuda touched in the launcher parent")
        for name in ("device_count", "is_available", "init", "current_device", "set_device"):
            setattr(torch.cuda, name, _forbidden)
    if args.backend == "nccl" and not args.dry_launch:
        visible = visible_gpu_count()
        if visible is not None and n > visible:
            print(f"bench.py: --gpus {n} but only {visible} GPU(s) are visible: one rank per GPU (RCCL refuses "
                  f"two ranks on one device)", file=sys.stderr, flush=True)
            return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("GSCAN_FORBID_HIP_IN_PARENT", None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0's stdout is drained by a thread so that the parent can watch EVERY child: the first rank that exits
    # non-zero ends the run (the others would otherwise sit in a rendezvous or a collective until torch's timeout)
    captured = []
    reader = threading.Thread(target=lambda: captured.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * n
    failed = None
    deadline_after_rank0 = None
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None and p.poll() is not None:
                codes[i] = p.returncode
                if p.returncode != 0 and failed is None:
                    failed = i
        if failed is not None:
            break
        if codes[0] is not None and deadline_after_rank0 is None:
            deadline_after_rank0 = time.monotonic() + 120     # a rank that outlives rank 0 by two minutes is stuck
        if deadline_after_rank0 is not None and time.monotonic() > deadline_after_rank0:
            break
        time.sleep(0.05)
    for i, p in enumerate(procs):
        if codes[i] is None:                                   # exactly the children this parent started
            p.kill()
            codes[i] = p.wait()
            if failed is None:
                failed = i
    reader.join(timeout=10)
    out = captured[0] if captured else b""
    for line in out.decode(errors="replace").splitlines():
        # ONE JSON line is the contract; anything else a library wrote to rank 0's stdout (gloo's connection notes) is
        # passed on to stderr
        print(line, file=sys.stdout if line.startswith("{") and failed is None else sys.stderr, flush=True)
    if failed is not None or any(codes):
        print(f"bench.py: rank exit codes {codes} (first failure: rank {failed})", file=sys.stderr, flush=True)
        return next((c for c in codes if c), 1) or 1
    return 0


def dry_launch(args, world: int, rank: int) -> None:
    """--dry-launch: the rank plumbing only (rendezvous, a barrier, one all-reduce), no model and no device: what the
    CPU test of the self-launcher runs with --backend gloo."""
    dist.init_process_group(backend=args.backend)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "backend": args.backend, "rank_sum": float(t.item())}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="rows per GPU (weak scaling: the global batch grows with N)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling (SURVEY.md 8d's secondary line): a fixed global batch, e.g. 2048, split "
                         "evenly over the N ranks; the line then says \"scaling\": \"strong\"")
    ap.add_argument("--target-length", type=int, default=20)
    ap.add_argument("--command-length", type=int, default=10)
    ap.add_argument("--workload", default="compositional", choices=["compositional", "target_length", "demo"])
    ap.add_argument("--auxiliary", action="store_true", help="S4: the GECA configuration (auxiliary target-position head)")
    ap.add_argument("--ragged", action="store_true", help="ragged command / target lengths instead of dense ones")
    ap.add_argument("--with-batcher", action="store_true",
                    help="also time the step fed by the dataset reader / batcher on a generated dataset file (SURVEY 8 f1)")
    ap.add_argument("--batcher-examples", type=int, default=100000)
    ap.add_argument("--min-warmup-steps", type=int, default=300,
                    help="untimed warm-up runs at least this many steps (0.15 s at S1), whatever --warmup says")
    ap.add_argument("--warmup-seconds", type=float, default=2.0,
                    help="untimed warm-up lasts at least this long: rank 0 times a short calibration run, derives a STEP "
                         "COUNT from it and broadcasts the count, so every rank still runs the same number of steps "
                         "(a fresh box needs ~1 s of work before its first timed window is representative)")
    ap.add_argument("--windows", type=int, default=4, help="extra timed windows of K steps for the spread (0 = none)")
    ap.add_argument("--attempts", type=int, default=3,
                    help="how often the timed windows are measured while the first window is > 3 %% off their median")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--rotate", type=int, default=4,
                    help="distinct resident batches the steps cycle through (1 = the same batch every step)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1: nccl = RCCL over xGMI (one rank per GPU); gloo = rehearsal "
                         "(gradients staged through the host, ranks may share a device)")
    ap.add_argument("--dry-launch", action="store_true", help="N > 1: exercise the rank plumbing only (no device)")
    ap.add_argument("--always-collective", action="store_true",
                    help="issue the data-parallel collectives even with ONE rank (needs the launcher environment: "
                         "WORLD_SIZE=1 RANK=0 MASTER_*): the RCCL communicator and its stream hand-over on a single GPU")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become one.  Nothing above this line has touched the device.
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.global_batch:
        if args.global_batch % args.gpus:
            raise SystemExit(f"--global-batch {args.global_batch} does not divide over --gpus {args.gpus}")
        args.batch = args.global_batch // args.gpus
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}, or without a launcher at all")
    if args.dry_launch:
        return dry_launch(args, world, rank)
    # ONE JSON line on stdout is the contract, and native libraries write there too (RCCL prints a version banner when its
    # first communicator comes up): from here on file descriptor 1 is stderr, and the line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the HIP device (the product path has no CPU fallback)")
    if args.backend == "gloo":
        local %= torch.cuda.device_count()                # rehearsal: ranks may share a device
    elif local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but {torch.cuda.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local)
    if world > 1 or args.always_collective:
        if world == 1:                                        # --always-collective without a launcher: a one-rank group
            for key, value in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"),
                               ("MASTER_PORT", str(29500 + os.getpid() % 2000))):
                os.environ.setdefault(key, value)
        # "nccl" IS RCCL on ROCm.  The process group carries the rendezvous, the barriers and the max-over-ranks of the
        # timings; the gradient all-reduce itself runs through the library's own communicator on the step's stream
        # (train.RcclCommunicator), created from this group.
        dist.init_process_group(backend=args.backend, **({"device_id": torch.device("cuda", local)}
                                                         if args.backend == "nccl" else {}))

    from multimodal_seq2seq_gscan_amd import _lib
    from multimodal_seq2seq_gscan_amd.config import model_kwargs
    from multimodal_seq2seq_gscan_amd.model import Model
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
    from multimodal_seq2seq_gscan_amd.train import TrainStep

    lib = _lib.load()
    cfg = model_kwargs(args.workload, auxiliary_task=args.auxiliary)
    grid = 4 if args.workload == "demo" else 6
    shape = Shape(batch=args.batch, grid=grid, channels=cfg["num_cnn_channels"],
                  input_vocab=cfg["input_vocabulary_size"], target_vocab=cfg["target_vocabulary_size"],
                  max_command=args.command_length, max_target=args.target_length, ragged=args.ragged)
    torch.manual_seed(42)                                 # identical initialisation on every rank (train.py:27)
    model = Model(**cfg).cuda()
    # --rotate K distinct batches, all resident in HBM before the timed region; the steps cycle through them
    batches = []
    for k in range(max(1, args.rotate)):
        b = {key: v.cuda() for key, v in make_batch(shape, seed=1234 + rank + 1000 * k).items()}
        b["cmd_lengths"] = b["cmd_lengths"].to(torch.int32)
        batches.append(b)
    batch = batches[0]
    taken = [0]

    def next_batch():
        taken[0] += 1
        return batches[taken[0] % len(batches)]

    step = TrainStep(model, learning_rate=1e-3, always_collective=args.always_collective)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed warm-up: W steps as asked, and at least --min-warmup-steps (ten steps are 5 ms, not enough for clocks and
    # allocator pools to settle: the first timed window used to be the slowest of the five).  A COUNT, not a duration:
    # every rank must run the same number of steps, each has a collective in it.
    # ... and a duration: on a fresh box 300 steps (0.15 s) left the first timed window 19 % above the median
    # (BENCH_r04: 0.5805 against 0.486 ms).  The duration target becomes a step COUNT on rank 0 (from a timed calibration
    # run every rank executes alike) and the count is broadcast: ranks never disagree on the number of collectives.
    calibrate = max(args.warmup, 20)
    for _ in range(calibrate):
        step(next_batch())
    fence()
    t0 = time.perf_counter()
    for _ in range(calibrate):
        step(next_batch())
    fence()
    per_step_s = (time.perf_counter() - t0) / calibrate
    warmup_steps = max(args.warmup, args.min_warmup_steps, min(100000, int(args.warmup_seconds / max(per_step_s, 1e-6)) + 1))
    if world > 1:
        count = torch.tensor([warmup_steps], dtype=torch.int64, device="cpu" if args.backend == "gloo" else "cuda")
        dist.broadcast(count, src=0)
        warmup_steps = int(count.item())
    for _ in range(warmup_steps):
        step(next_batch())
    warmup_steps += 2 * calibrate
    tdev = "cpu" if args.backend == "gloo" else "cuda"

    def timed_window():
        """exactly K steps between barrier + synchronize pairs; the maximum over ranks, in seconds"""
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = step(next_batch())
        fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=tdev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), last

    # The contract's number is the FIRST window after the warm-up; `value` is the median of 1 + --windows such windows (one
    # window is 25 ms of wall clock: a single host or device stall of a few milliseconds would otherwise be the number of
    # the round).  The two must agree: when the first window is more than 3 % off the median, the whole measurement is
    # repeated (up to --attempts times, every rank alike: the decision is taken on the all-reduced times); if it still
    # is, `value` is the FIRST window (what --steps/--warmup literally describe) and the line says so.
    attempts = 0
    while True:
        attempts += 1
        elapsed, out = timed_window()
        windows = [1e3 * elapsed / args.steps]
        for _ in range(args.windows):
            windows.append(1e3 * timed_window()[0] / args.steps)
        median_ms = sorted(windows)[len(windows) // 2]
        first_off = abs(windows[0] - median_ms) / median_ms
        if first_off <= 0.03 or attempts >= max(1, args.attempts) or args.windows == 0:
            break
        if rank == 0:
            print(f"bench.py: first window {windows[0]:.4f} ms is {100 * first_off:.1f} % off the median {median_ms:.4f} ms: "
                  f"measuring again (attempt {attempts + 1})", file=sys.stderr, flush=True)
    loss = float(out["loss"].item())

    # second pass over the same K steps with HIP-event probes around each kernel family
    probe_step = step
    for _ in range(3):
        probe_step(next_batch())
    torch.cuda.synchronize()
    lib.gscan_probe_reset()
    lib.gscan_probe_enable(1)
    for _ in range(args.steps):
        probe_step(next_batch())
    torch.cuda.synchronize()
    lib.gscan_probe_enable(0)
    families = {}
    for name in PROBES:
        ms, xfl, fl, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        _lib.check(lib.gscan_probe_read(name.encode(), C.byref(ms), C.byref(xfl), C.byref(fl), C.byref(n)),
                   "gscan_probe_read")
        if n.value:
            per_s = 1.0 / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
            families[name] = {"ms_per_step": ms.value / args.steps, "launches_per_step": n.value / args.steps,
                              "avg_us": 1e3 * ms.value / n.value,
                              "tflops": fl.value * per_s,                    # ALGORITHMIC flops / time
                              # multiply-adds the launches EXECUTE (GEMMs: 2 M N K incl. the gate-image products; decoder:
                              # DESIGN.md 4.1's count); None where that depends on the data (the input-sparse convolutions)
                              "executed_tflops": xfl.value * per_s if xfl.value > 0 else None,
                              "algorithmic_gflop_per_launch": fl.value / n.value / 1e9}
    fence()

    batcher = None
    if args.with_batcher and rank == 0 and world == 1:
        batcher = batcher_leg(args, cfg, model.parameter_count)
    if rank == 0:
        B, L, T = args.batch, args.command_length, args.target_length
        # `value`: the MEDIAN of the timed windows (each is exactly K steps between barrier + synchronize pairs, maximum
        # over ranks).  One window is 25 ms of wall clock and a single host or device stall of a few milliseconds —
        # seen twice in some sixty runs of a round: 0.89 ms against 0.49 — would otherwise be the number of the round;
        # the first window and the spread are reported next to it.
        first_window_ms = 1e3 * elapsed / args.steps
        first_window_ok = first_off <= 0.03
        if first_window_ok:
            elapsed = 1e-3 * sorted(windows)[len(windows) // 2] * args.steps
        ex_per_s = world * B * args.steps / elapsed
        # The roofline block prices the GEMM family (the conv + LSTM + projection products north_star's MFMA target
        # names).  So that the line cannot flatter: `families_ranked` lists EVERY family by time per step with its
        # own algorithmic TFLOP/s and fraction of the same peak, the two persistent decoder kernels summed as one
        # family (together they are the largest consumer).
        merged = dict(families)
        if "decoder_forward" in merged and "decoder_backward" in merged:
            f, b = merged.pop("decoder_forward"), merged.pop("decoder_backward")
            ms = f["ms_per_step"] + b["ms_per_step"]
            alg = f["tflops"] * f["ms_per_step"] + b["tflops"] * b["ms_per_step"]      # TFLOP/s x ms = GFLOP per step
            xalg = (f["executed_tflops"] or 0.0) * f["ms_per_step"] + (b["executed_tflops"] or 0.0) * b["ms_per_step"]
            n = f["launches_per_step"] + b["launches_per_step"]
            merged["decoder_pair"] = {"ms_per_step": ms, "launches_per_step": n, "tflops": alg / ms if ms > 0 else 0.0,
                                      "executed_tflops": xalg / ms if ms > 0 else 0.0, "avg_us": 1e3 * ms / n,
                                      "algorithmic_gflop_per_launch": alg / n}
        ranked = [{"family": k, "ms_per_step": round(v["ms_per_step"], 4), "launches_per_step": round(v["launches_per_step"], 2),
                   "algorithmic_tflops": round(v["tflops"], 3), "frac_of_fp32_peak": round(v["tflops"] / PEAK_FP32_TFLOPS, 4)}
                  for k, v in sorted(merged.items(), key=lambda kv: -kv[1]["ms_per_step"])]
        mflop = algorithmic_mflop_per_example(cfg, grid, L, T)
        # `roofline` = the family with the LARGEST time per step (the decoder's two recurrences count as one family);
        # `roofline_gemm` = the grouped-GEMM family beside it (the conv + LSTM + projection products north_star's MFMA
        # target names), whichever of the two is not already in `roofline`.
        dominant = max(merged, key=lambda k: merged[k]["ms_per_step"])
        pmc_workload = args.workload if not args.auxiliary and T == 20 else "other"
        pmc_names = {"gemm": ["gemm"], "decoder_pair": ["decoder_fwd_kernel", "decoder_bwd_kernel"]}

        def roofline_block(name):
            d = merged[name]
            traffic, traffic_source, traffic_all = pmc_traffic("gemm", pmc_workload)
            if name == "decoder_pair":        # bytes per launch of the pair = the mean of its two kernels
                pair = [traffic_all.get(k) for k in pmc_names["decoder_pair"]]
                traffic = round(sum(pair) / 2) if all(v is not None for v in pair) else None
            elif name != "gemm":          # a family without its own PMC figure: null (the per-kernel table is beside it)
                traffic = None
            return {"bound": "mfma", "kernel": name, "achieved": round(d["tflops"], 3), "peak": PEAK_FP32_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(d["tflops"] / PEAK_FP32_TFLOPS, 4), "traffic": traffic,
                    "traffic_source": traffic_source, "traffic_by_kernel": traffic_all,
                    "ms_per_step": round(d["ms_per_step"], 4), "launches_per_step": round(d["launches_per_step"], 2),
                    "avg_launch_us": round(d["avg_us"], 2),
                    "algorithmic_gflop_per_launch": round(d["algorithmic_gflop_per_launch"], 4),
                    "executed_over_algorithmic": round(d["executed_tflops"] / d["tflops"], 3) if d["tflops"] > 0 and d["executed_tflops"] else None,
                    "note": "achieved = ALGORITHMIC flops of the family's launches (SURVEY.md 8d; per-family formulas in "
                            "DESIGN.md 4) / their HIP-event time on the launch streams; fp32 matrix peak = fp32 vector "
                            "peak on gfx950; traffic = HBM-side bytes per launch from the rocprofv3 PMC passes named in "
                            "traffic_source, null when that profile is stale"}
        windows_sorted = sorted(windows)
        label = {"compositional": "S4 GECA-like (auxiliary head)" if args.auxiliary else "S1 compositional",
                 "target_length": "S3 target_length", "demo": "S0 demo"}[args.workload]
        result = {
            "metric": "training examples/sec (forward+backward) on compositional_splits, 1/2/4/8 GPUs",
            "value": round(ex_per_s, 1), "unit": "examples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "ms_per_step_first_window": round(first_window_ms, 4),
            "ms_per_step_windows": {"n": len(windows), "min": round(windows_sorted[0], 4),
                                    "median": round(windows_sorted[len(windows) // 2], 4),
                                    "max": round(windows_sorted[-1], 4)},
            "config": {"workload": f"{label}: {B} rows/GPU, {grid}x{grid}x{cfg['num_cnn_channels']} grid, "
                                   f"k={cfg['cnn_kernel_size']}, hidden {cfg['decoder_hidden_size']}, L={L}, T={T} "
                                   f"{'ragged' if args.ragged else 'dense'}, "
                                   f"dropout {cfg['encoder_dropout_p']}/{cfg['decoder_dropout_p']}/{cfg['cnn_dropout_p']}, "
                                   f"conditional attention{', auxiliary head' if args.auxiliary else ''}, Adam+LR step included",
                       "global_batch": world * B, "parallelism": f"dp{world}", "parameters": model.parameter_count,
                       "launch": "eager (forward on the caller's stream, backward on 3 streams)",
                       "resident_batches": len(batches),
                       "value_window": ("the one timed window" if len(windows) == 1 else
                                        "median of the timed windows of K steps each" if first_window_ok else
                                        "FIRST window: it stayed > 3 % off the median of the windows over every attempt"),
                       "first_window_off_median": round(first_off, 4), "measurement_attempts": attempts,
                       # 1 = the register/LDS-resident decoder kernels, 0 = the streaming ones (gscan_decoder_kernel_family)
                       "decoder_kernels": {1: "resident (csrc/decoder.hip)", 0: "streaming (csrc/decoder_any.hip)"}.get(
                           int(lib.gscan_decoder_kernel_family(C.byref(model._dims(B, L, T, grid)))), "unsupported"),
                       "warmup_steps_run": warmup_steps, "warmup_seconds_target": args.warmup_seconds,
                       "gradient_exchange": (None if not step.exchange.collective else
                                             "gscan_allreduce_f32: RCCL on the step's stream" if step.exchange.comm is not None
                                             else f"torch.distributed {args.backend}"),
                       "dp_buckets": step.exchange.buckets if step.exchange.collective else None,
                       # ranks of the communicator the step's all-reduce runs on, as RCCL itself reports them
                       "rccl_nranks": step.exchange.comm.nranks if step.exchange.comm is not None else None},
            "roofline": roofline_block(dominant),
            "roofline_gemm": roofline_block("gemm") if dominant != "gemm" and "gemm" in merged else None,
            # since round 6 the two largest families are within a few per cent of each other: whichever is not `roofline`
            "roofline_decoder_pair": roofline_block("decoder_pair") if dominant != "decoder_pair" and "decoder_pair" in merged else None,
            "step_hbm_traffic": step_traffic(pmc_workload, mflop, B),
            "families_ranked": ranked,
            "kernel_families": {k: {kk: (None if vv is None else round(vv, 3)) for kk, vv in v.items()} for k, v in families.items()},
            "step_algorithmic_tflops": round(ex_per_s * mflop / 1e6, 3),
            "final_loss": round(loss, 4),
        }
        if batcher is not None:
            result["with_batcher"] = batcher
        if world == 1 and args.cpu_seconds > 0:
            result["cpu_baseline"] = cpu_baseline(cfg, shape, args.cpu_seconds)
        else:
            result["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    step.close()                       # the library's own RCCL communicator goes before the process group
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
