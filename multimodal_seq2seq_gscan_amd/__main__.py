"""Command line of the reference (`python -m seq2seq --mode=train ...`, seq2seq/__main__.py:21-167).

Flag names, defaults and the paired on/off switches are the reference's; two flags are additive:
`--synthetic_data` trains on seeded synthetic batches of the data set's shape (no dataset file ships
with the reference checkout, .MISSING_LARGE_BLOBS) and `--synthetic_batches` bounds that run.
Data-parallel runs are launched with `python -m torch.distributed.run --nproc-per-node N -m seq2seq ...`;
each rank then trains on its shard of every global batch (train.py in this package).
"""
from __future__ import annotations

import argparse
import logging
import os

import torch

logger = logging.getLogger(__name__)

# The reference's flag surface (seq2seq/__main__.py:21-102) as a table: (name, type, default).
_VALUE_FLAGS = [
    ("mode", str, None), ("output_directory", str, "output"), ("resume_from_file", str, ""),
    ("split", str, "test"), ("data_directory", str, "data/uniform_dataset"),
    ("input_vocab_path", str, "training_input_vocab.txt"), ("target_vocab_path", str, "training_target_vocab.txt"),
    ("training_batch_size", int, 50), ("k", int, 0), ("test_batch_size", int, 1),
    ("max_training_examples", int, None), ("learning_rate", float, 0.001), ("lr_decay", float, 0.9),
    ("lr_decay_steps", float, 20000), ("adam_beta_1", float, 0.9), ("adam_beta_2", float, 0.999),
    ("print_every", int, 100), ("evaluate_every", int, 1000), ("max_training_iterations", int, 100000),
    ("weight_target_loss", float, 0.3), ("max_testing_examples", int, None), ("splits", str, "test"),
    ("max_decoding_steps", int, 30), ("output_file_name", str, "predict.json"),
    ("cnn_hidden_num_channels", int, 50), ("cnn_kernel_size", int, 7), ("cnn_dropout_p", float, 0.1),
    ("embedding_dimension", int, 25), ("num_encoder_layers", int, 1), ("encoder_hidden_size", int, 100),
    ("encoder_dropout_p", float, 0.3), ("num_decoder_layers", int, 1), ("decoder_dropout_p", float, 0.3),
    ("decoder_hidden_size", int, 100), ("seed", int, 42),
    # additive: synthetic data (no dataset file ships with the reference checkout)
    ("synthetic_batches", int, 1000),
    # additive: order every window of this many batches by target length after the shuffle (0 = the reference's order)
    ("length_bucket_batches", int, 0),
]
# paired switches: (destination, flag that sets True, flag that sets False, default)
_SWITCHES = [
    ("generate_vocabularies", "generate_vocabularies", "load_vocabularies", False),
    ("simple_situation_representation", "simple_situation_representation", "image_situation_representation", True),
    ("auxiliary_task", "auxiliary_task", "no_auxiliary_task", False),
    ("encoder_bidirectional", "encoder_bidirectional", "encoder_unidirectional", True),
    ("conditional_attention", "conditional_attention", "no_conditional_attention", True),
    ("synthetic_data", "synthetic_data", "dataset_file", False),
]


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="gSCAN seq2seq training on MI355X (flags of the reference CLI)")
    for name, kind, default in _VALUE_FLAGS:
        p.add_argument("--" + name, type=kind, default=default, required=(name == "mode"))
    p.add_argument("--attention_type", type=str, default="bahdanau", choices=["bahdanau", "luong"])
    for dest, on, off, default in _SWITCHES:
        p.add_argument("--" + on, dest=dest, action="store_true", default=default)
        p.add_argument("--" + off, dest=dest, action="store_false")
    return p


parser = build_parser()


def _init_distributed():
    """One process per GPU when launched through torch.distributed.run; single process otherwise."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # GSCAN_DIST_BACKEND=gloo: rehearsal of the rank plumbing with the ranks sharing devices (gradients staged through the
    # host; RCCL wants a device per rank)
    backend = os.environ.get("GSCAN_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local % max(1, torch.cuda.device_count()) if backend == "gloo" else local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend=backend)    # "nccl" is RCCL on ROCm
    return rank, world


def _synthetic_batches(flags, rank: int, world: int):
    from .synthetic import Shape, make_batch
    from .train import shard_batch
    shape = Shape(batch=flags["training_batch_size"], ragged=True)
    for i in range(flags["synthetic_batches"]):
        cpu = make_batch(shape, seed=flags["seed"] * 100003 + i)       # same global batch on every rank
        yield {k: v.cuda(non_blocking=True) for k, v in shard_batch(cpu, rank, world).items()}


def _test_on_dataset(flags) -> None:
    """seq2seq/__main__.py:124-163 + predict.py:17-54: greedy-decode every example of each split of a gSCAN dataset
    file and write <split>_<output_file_name> with the reference's records (words, not ids)."""
    import json
    from .config import model_kwargs
    from .dataset import GroundedScanDataset
    from .model import Model
    from .predict import predict_and_save
    data_path = os.path.join(flags["data_directory"], "dataset.txt")
    assert os.path.exists(os.path.join(flags["data_directory"], flags["input_vocab_path"])) and os.path.exists(
        os.path.join(flags["data_directory"], flags["target_vocab_path"])), \
        "No vocabs found at {} and {}".format(flags["input_vocab_path"], flags["target_vocab_path"])
    for split in flags["splits"].split(","):
        logger.info("Loading {} dataset split...".format(split))
        test_set = GroundedScanDataset(data_path, flags["data_directory"], split=split,
                                       input_vocabulary_file=flags["input_vocab_path"],
                                       target_vocabulary_file=flags["target_vocab_path"], generate_vocabulary=False,
                                       k=flags["k"])
        test_set.read_dataset(max_examples=None)
        logger.info("Done Loading {} dataset split.".format(flags["split"]))
        logger.info("  Loaded {} examples.".format(test_set.num_examples))
        logger.info("  Input vocabulary size: {}".format(test_set.input_vocabulary_size))
        logger.info("  Output vocabulary size: {}".format(test_set.target_vocabulary_size))
        cfg = model_kwargs("compositional")
        cfg.update({k: flags[k] for k in cfg if k in flags})
        cfg.update(input_vocabulary_size=test_set.input_vocabulary_size,
                   target_vocabulary_size=test_set.target_vocabulary_size, num_cnn_channels=test_set.image_channels,
                   input_padding_idx=test_set.input_vocabulary.pad_idx, target_pad_idx=test_set.target_vocabulary.pad_idx,
                   target_eos_idx=test_set.target_vocabulary.eos_idx)
        model = Model(**cfg).cuda()
        assert os.path.isfile(flags["resume_from_file"]), "No checkpoint found at {}".format(flags["resume_from_file"])
        logger.info("Loading checkpoint from file at '{}'".format(flags["resume_from_file"]))
        model.load_model(flags["resume_from_file"])
        logger.info("Loaded checkpoint '{}' (iter {})".format(flags["resume_from_file"], model.trained_iterations))
        output_file_path = os.path.join(flags["output_directory"], "_".join([split, flags["output_file_name"]]))
        predict_and_save(test_set, model, output_file_path, flags["max_decoding_steps"],
                         max_testing_examples=flags["max_testing_examples"])
        logger.info("Saved predictions to {}".format(output_file_path))


def main(flags):
    logging.basicConfig(format="%(asctime)-15s %(message)s", level=logging.DEBUG, datefmt="%Y-%m-%d %H:%M")
    for argument, value in flags.items():
        logger.info("{}: {}".format(argument, value))
    if not os.path.exists(flags["output_directory"]):
        os.makedirs(flags["output_directory"], exist_ok=True)
    if not flags["simple_situation_representation"]:
        raise NotImplementedError("Full RGB input image not implemented. Implement or set "
                                  "--simple_situation_representation")
    if flags["generate_vocabularies"]:
        assert flags["input_vocab_path"] and flags["target_vocab_path"], "Please specify paths to vocabularies to save."
    if flags["test_batch_size"] > 1:
        raise NotImplementedError("Test batch size larger than 1 not implemented.")

    if flags["mode"] == "train":
        from .config import model_kwargs
        from .model import Model
        from .train import train
        if not torch.cuda.is_available():
            raise RuntimeError("training runs on the HIP device only (no CPU fallback in this package)")
        rank, world = _init_distributed()
        torch.manual_seed(flags["seed"])                      # train.py:27 (identical init on every rank)
        if not flags["synthetic_data"]:
            from .train import train_on_dataset
            train_on_dataset(data_path=os.path.join(flags["data_directory"], "dataset.txt"), rank=rank,
                             world_size=world, **flags)
            return
        cfg = model_kwargs("compositional")
        cfg.update({k: flags[k] for k in cfg if k in flags})
        model = Model(**cfg).cuda()
        optimizer_state = None
        if flags["resume_from_file"]:
            assert os.path.isfile(flags["resume_from_file"]), "No checkpoint found at {}".format(flags["resume_from_file"])
            optimizer_state = model.load_model(flags["resume_from_file"])      # train.py:81-82
        step = train(_synthetic_batches(flags, rank, world), model, flags["max_training_iterations"],
                     print_every=flags["print_every"], weight_target_loss=flags["weight_target_loss"], rank=rank,
                     optimizer_state_dict=optimizer_state,
                     learning_rate=flags["learning_rate"], adam_beta_1=flags["adam_beta_1"],
                     adam_beta_2=flags["adam_beta_2"], lr_decay=flags["lr_decay"],
                     lr_decay_steps=flags["lr_decay_steps"])
        if rank == 0:
            model.save_checkpoint("checkpoint.pth.tar", is_best=False, optimizer_state_dict=step.optimizer.state_dict())
        logger.info("Finished training.")
    elif flags["mode"] == "test":
        # seq2seq/__main__.py:124-163 + predict.py:17-54: greedy-decode every example of each split and write
        # <split>_<output_file_name> with the reference's record schema.  Synthetic data has no vocabulary, so the
        # "input" / "prediction" / "target" fields hold token ids instead of words.
        import json
        from .config import model_kwargs
        from .model import Model
        from .predict import predict, sequence_accuracy
        from .synthetic import Shape, make_batch
        if not torch.cuda.is_available():
            raise RuntimeError("greedy decoding runs on the HIP device only (no CPU fallback in this package)")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        if not flags["synthetic_data"]:
            _test_on_dataset(flags)
            return
        assert os.path.isfile(flags["resume_from_file"]), "No checkpoint found at {}".format(flags["resume_from_file"])
        cfg = model_kwargs("compositional")
        cfg.update({k: flags[k] for k in cfg if k in flags})
        model = Model(**cfg).cuda()
        logger.info("Loading checkpoint from file at '{}'".format(flags["resume_from_file"]))
        model.load_model(flags["resume_from_file"])
        logger.info("Loaded checkpoint '{}' (iter {})".format(flags["resume_from_file"], model.trained_iterations))
        n_examples = flags["max_testing_examples"] or 64
        for split_index, split in enumerate(flags["splits"].split(",")):
            def iterator():
                done = 0
                while done < n_examples:
                    n = min(256, n_examples - done)
                    b = make_batch(Shape(batch=n, ragged=True), seed=flags["seed"] * 7919 + 31 * split_index + done)
                    yield (b["commands"].cuda(), b["cmd_lengths"].tolist(), [None] * n, b["world"].cuda(), [None] * n,
                           b["targets"].cuda(), b["tgt_lengths"].tolist(), None, b["target_positions"].cuda())
                    done += n
            output = []
            for (inp, derivation, situation, out_seq, tgt, aw_c, aw_s, pos_acc) in predict(
                    iterator(), model=model, max_decoding_steps=flags["max_decoding_steps"],
                    pad_idx=cfg["target_pad_idx"], sos_idx=1, eos_idx=cfg["target_eos_idx"]):
                accuracy = sequence_accuracy(out_seq, tgt[0].tolist()[1:-1])
                output.append({"input": inp[0].tolist()[1:-1], "prediction": out_seq, "derivation": derivation,
                               "target": tgt[0].tolist()[1:-1], "situation": situation,
                               "attention_weights_input": aw_c, "attention_weights_situation": aw_s,
                               "accuracy": accuracy, "exact_match": True if accuracy == 100 else False,
                               "position_accuracy": pos_acc})
            output_file_path = os.path.join(flags["output_directory"], "_".join([split, flags["output_file_name"]]))
            with open(output_file_path, mode="w") as outfile:
                json.dump(output, outfile, indent=4)
            logger.info("Wrote predictions for {} examples.".format(len(output)))
            logger.info("Saved predictions to {}".format(output_file_path))
    elif flags["mode"] == "predict":
        raise NotImplementedError()
    else:
        raise ValueError("Wrong value for parameters --mode ({}).".format(flags["mode"]))


if __name__ == "__main__":
    main(flags=vars(parser.parse_args()))
