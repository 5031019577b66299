"""Seeded synthetic gSCAN batches in the reference's batch contract.

The reference's ``GroundedScanDataset.get_data_iterator`` (seq2seq/gSCAN_dataset.py:184-231)
yields ``(input_batch [B,L] i64, input_lengths, derivation, situation_batch [B,G,G,C] f32,
situation_repr, target_batch [B,T] i64, target_lengths, agent_positions, target_positions [B])``.
No dataset file ships with the reference checkout, so benchmarks and parity tests draw
batches of the same shape, dtype and value distribution from a seeded generator
(SURVEY.md §8d): tokens PAD=0/SOS=1/EOS=2/words>=3 (gSCAN_dataset.py:22-32), world cells
``[size one-hot(4) | shape | colour | agent | direction one-hot(4)]`` with values in {0,1}
(GroundedScan/gym_minigrid/minigrid.py:380-399).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import torch

PAD, SOS, EOS = 0, 1, 2


@dataclass(frozen=True)
class Shape:
    """Dimensions of one workload (names follow the reference's flags)."""
    batch: int
    grid: int = 6
    channels: int = 16
    input_vocab: int = 21
    target_vocab: int = 9
    max_command: int = 10
    max_target: int = 20
    ragged: bool = False


# The workloads named in BASELINE.md §3.
S0_DEMO = Shape(batch=4, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7, max_target=10,
                ragged=True)
S1_COMPOSITIONAL = Shape(batch=256)
S3_TARGET_LENGTH = Shape(batch=256, input_vocab=17, target_vocab=8, max_target=120)


def _sequences(gen: torch.Generator, B: int, vocab: int, max_len: int, min_len: int, ragged: bool):
    """Rows ``[SOS, w..., EOS, PAD...]``; lengths count SOS and EOS."""
    if ragged:
        lengths = torch.randint(min_len, max_len + 1, (B,), generator=gen)
        lengths[0] = max_len                      # the batch is padded to its longest row
    else:
        lengths = torch.full((B,), max_len, dtype=torch.long)
    words = torch.randint(3, vocab, (B, max_len), generator=gen)
    pos = torch.arange(max_len).unsqueeze(0)
    seq = torch.where(pos < (lengths - 1).unsqueeze(1), words, torch.full_like(words, PAD))
    seq[:, 0] = SOS
    seq.scatter_(1, (lengths - 1).unsqueeze(1), EOS)
    return seq, lengths


def _worlds(gen: torch.Generator, B: int, G: int, C: int) -> torch.Tensor:
    """Sparse {0,1} world tensors: one agent cell plus 1..12 objects on distinct cells."""
    n_attr = C - 5                                 # size(4) + shape + colour one-hots
    n_shape = max(1, (n_attr - 4) // 2)
    n_colour = max(1, n_attr - 4 - n_shape)
    world = torch.zeros(B, G * G, C)
    for b in range(B):
        cells = torch.randperm(G * G, generator=gen)
        n_obj = int(torch.randint(1, min(12, G * G - 1) + 1, (1,), generator=gen))
        agent = int(cells[0])
        world[b, agent, C - 5] = 1.0
        world[b, agent, C - 4 + int(torch.randint(0, 4, (1,), generator=gen))] = 1.0
        for cell in cells[1:1 + n_obj].tolist():
            world[b, cell, int(torch.randint(0, 4, (1,), generator=gen))] = 1.0
            world[b, cell, 4 + int(torch.randint(0, n_shape, (1,), generator=gen))] = 1.0
            world[b, cell, 4 + n_shape + int(torch.randint(0, n_colour, (1,), generator=gen))] = 1.0
    return world.view(B, G, G, C)


def make_batch(shape: Shape, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """One CPU batch.  Keys: commands, cmd_lengths, world, targets, tgt_lengths, target_positions."""
    gen = torch.Generator().manual_seed(seed)
    commands, cmd_lengths = _sequences(gen, shape.batch, shape.input_vocab, shape.max_command,
                                       min(5, shape.max_command), shape.ragged)
    targets, tgt_lengths = _sequences(gen, shape.batch, shape.target_vocab, shape.max_target,
                                      min(4, shape.max_target), shape.ragged)
    world = _worlds(gen, shape.batch, shape.grid, shape.channels)
    positions = torch.randint(0, shape.grid * shape.grid, (shape.batch,), generator=gen)
    return {"commands": commands, "cmd_lengths": cmd_lengths, "world": world, "targets": targets,
            "tgt_lengths": tgt_lengths, "target_positions": positions}


# ------------------------------------------------------------------------------------------------------------------
# A dataset FILE of the same distribution, in the reference's format (GroundedScan/dataset.py:487-514 reads it;
# the situation dictionaries follow Situation.to_representation, world.py:269-281): input for the reader / batcher
# (dataset.py here) when no gSCAN download is at hand — tests and `bench.py --with-batcher`.
# ------------------------------------------------------------------------------------------------------------------
_COMMAND_WORDS = ["walk", "push", "pull", "to", "a", "the", "red", "green", "blue", "yellow", "big", "small", "circle",
                  "square", "cylinder", "cautiously", "hesitantly", "while", "spinning", "zigzagging"]
_ACTIONS = ["walk", "turn left", "turn right", "push", "pull", "stay"]
_SHAPES, _COLOURS = ["circle", "square", "cylinder"], ["red", "green", "blue", "yellow"]


def write_dataset_file(path: str, examples: Dict[str, int], shape: Shape = Shape(batch=1), seed: int = 0) -> None:
    """Write {"examples": {split: [...]}} with examples[split] random examples per split: commands of
    4..max_command-2 words, targets of 2..max_target-2 actions (ragged), grids of shape.grid cells with one agent
    and 1..12 objects (4 sizes + 3 shapes + 4 colours -> 11 attributes, 16 channels)."""
    import json
    import random
    rng = random.Random(seed)
    G = shape.grid
    out = {"grid_size": G, "type_grammar": "adverb", "min_object_size": 1, "max_object_size": 4, "max_recursion": 2,
           "percentage_train": 0.8, "examples": {}}
    for split, count in examples.items():
        rows = []
        for _ in range(count):
            cells = rng.sample(range(G * G), min(G * G, 1 + rng.randint(1, 12)))
            placed = {}
            for j, cell in enumerate(cells[1:]):
                size, sh, co = rng.randint(1, 4), rng.randrange(3), rng.randrange(4)
                vector = ["0"] * 11
                vector[size - 1], vector[4 + sh], vector[7 + co] = "1", "1", "1"
                placed[str(j)] = {"vector": "".join(vector),
                                  "position": {"row": str(cell // G), "column": str(cell % G)},
                                  "object": {"shape": _SHAPES[sh], "color": _COLOURS[co], "size": str(size)}}
            target = placed["0"]
            situation = {"grid_size": G, "agent_position": {"row": str(cells[0] // G), "column": str(cells[0] % G)},
                         "agent_direction": rng.randrange(4), "target_object": target, "distance_to_target": "1",
                         "direction_to_target": "n", "placed_objects": placed, "carrying_object": None}
            command = [rng.choice(_COMMAND_WORDS) for _ in range(rng.randint(min(4, shape.max_command - 2),
                                                                             shape.max_command - 2))]
            actions = [rng.choice(_ACTIONS) for _ in range(rng.randint(min(2, shape.max_target - 2),
                                                                       shape.max_target - 2))]
            rows.append({"command": ",".join(command), "meaning": ",".join(command), "derivation": "",
                         "situation": situation, "target_commands": ",".join(actions), "verb_in_command": command[0],
                         "manner": "", "referred_target": ""})
        out["examples"][split] = rows
    with open(path, "w") as f:
        json.dump(out, f)
