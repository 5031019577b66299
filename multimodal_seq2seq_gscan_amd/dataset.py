"""gSCAN dataset reader and batcher without the simulator.

The reference reads `data/<split>/dataset.txt` through `GroundedScan.load_dataset_from_file`
(GroundedScan/dataset.py:487-514), re-plays every situation in its grid-world simulator to obtain the grid tensor
(`get_examples_with_image`, dataset.py:137-163 -> `Grid.encode`, gym_minigrid/minigrid.py:380-399), keeps every
example as separate device tensors and concatenates them per batch (seq2seq/gSCAN_dataset.py:184-278).

Here the JSON is read directly.  What the simulator would draw is already in the file: each placed object carries
its attribute vector (`"vector": "0001010100"` = sizes | shapes + colours one-hot, world.py:415-434) and its cell,
and the agent cell / direction are plain numbers, so the `[G, G, C]` tensor of `Grid.encode` — object vector, then
one agent bit, then a one-hot of the four directions, indexed `[row, column, :]` — is filled without `gym`.
Examples are packed ONCE into contiguous host arrays (tokens int64 padded to the split's longest sequence, grids
uint8: 576 bytes per 6x6x16 example).  A batch is gathered straight into ONE pinned slab of a small ring
(`BatchStager`), crosses PCIe as ONE asynchronous copy on a copy stream that runs a batch ahead of the consumer, and
its tensors are views of the device slab.  The world stays uint8 all the way: the convolution kernels widen it in
registers (csrc/conv.hip), so 576 B per example move over PCIe and through HBM instead of 2 304.
`batches()` yields the dictionaries `TrainStep` consumes; `get_data_iterator()` yields the reference's 9-tuple
(gSCAN_dataset.py:229-231), with a float32 world unless asked otherwise.
"""
from __future__ import annotations

import json
import logging
import os
import random
from collections import Counter, defaultdict
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

logger = logging.getLogger(__name__)


class Vocabulary:
    """seq2seq/gSCAN_dataset.py:17-102: <PAD>=0, <SOS>=1, <EOS>=2, then words in order of first appearance; unknown
    words map to <PAD>.  Same JSON on disk (`to_dict`)."""

    def __init__(self, sos_token="<SOS>", eos_token="<EOS>", pad_token="<PAD>"):
        self.sos_token, self.eos_token, self.pad_token = sos_token, eos_token, pad_token
        self._idx_to_word = [pad_token, sos_token, eos_token]
        self._word_to_idx = defaultdict(int)
        self._word_to_idx[pad_token] = 0
        self._word_to_idx[sos_token] = 1
        self._word_to_idx[eos_token] = 2
        self._word_frequencies = Counter()

    def word_to_idx(self, word: str) -> int:
        return self._word_to_idx.get(word, 0)

    def idx_to_word(self, idx: int) -> str:
        return self._idx_to_word[idx]

    def contains_word(self, word: str) -> bool:
        return self.word_to_idx(word) != 0

    def add_sentence(self, sentence: List[str]) -> None:
        for word in sentence:
            if word not in self._word_to_idx:
                self._word_to_idx[word] = self.size
                self._idx_to_word.append(word)
            self._word_frequencies[word] += 1

    def most_common(self, n=10):
        return self._word_frequencies.most_common(n=n)

    pad_idx = property(lambda self: self.word_to_idx(self.pad_token))
    sos_idx = property(lambda self: self.word_to_idx(self.sos_token))
    eos_idx = property(lambda self: self.word_to_idx(self.eos_token))
    size = property(lambda self: len(self._idx_to_word))

    @classmethod
    def load(cls, path: str) -> "Vocabulary":
        assert os.path.exists(path), "Trying to load a vocabulary from a non-existing file {}".format(path)
        with open(path, "r") as infile:
            data = json.load(infile)
        vocab = cls(sos_token=data["sos_token"], eos_token=data["eos_token"], pad_token=data["pad_token"])
        vocab._idx_to_word = list(data["idx_to_word"])
        vocab._word_to_idx = defaultdict(int)
        for word, idx in data["word_to_idx"].items():
            vocab._word_to_idx[word] = idx
        vocab._word_frequencies = Counter(data["word_frequencies"])
        return vocab

    def to_dict(self) -> dict:
        return {"sos_token": self.sos_token, "eos_token": self.eos_token, "pad_token": self.pad_token,
                "idx_to_word": self._idx_to_word, "word_to_idx": dict(self._word_to_idx),
                "word_frequencies": dict(self._word_frequencies)}

    def save(self, path: str) -> str:
        with open(path, "w") as outfile:
            json.dump(self.to_dict(), outfile, indent=4)
        return path


def parse_command_repr(command_repr: str) -> List[str]:
    """GroundedScan/dataset.py `parse_command_repr`: commands are comma-joined words."""
    return command_repr.split(",")


def encode_situation(situation: dict) -> np.ndarray:
    """`Grid.encode` (gym_minigrid/minigrid.py:380-399) from the situation dictionary of the data file:
    uint8 [G, G, C], C = len(object vector) + 1 + 4, indexed [row, column, :]."""
    G = int(situation["grid_size"])
    objects = situation["placed_objects"]
    n_attr = len(next(iter(objects.values()))["vector"]) if objects else len(situation["target_object"]["vector"])
    grid = np.zeros((G, G, n_attr + 5), dtype=np.uint8)
    for obj in objects.values():
        row, col = int(obj["position"]["row"]), int(obj["position"]["column"])
        grid[row, col, :n_attr] = np.frombuffer(obj["vector"].encode("ascii"), dtype=np.uint8) - ord("0")
    row, col = int(situation["agent_position"]["row"]), int(situation["agent_position"]["column"])
    grid[row, col, n_attr] = 1
    grid[row, col, n_attr + 1 + int(situation["agent_direction"])] = 1
    return grid


def load_examples(path_to_data: str, k: int = 0) -> Dict[str, List[dict]]:
    """The `examples` of a dataset file by split, with `k` random examples of the `adverb_1` split moved into
    train AND dev (GroundedScan/dataset.py:499-510; `random.sample`, unseeded there as well)."""
    with open(path_to_data, "r") as infile:
        all_data = json.load(infile)
    splits: Dict[str, List[dict]] = defaultdict(list)
    for split, examples in all_data["examples"].items():
        moved = set(random.sample(range(len(examples)), k=k)) if split == "adverb_1" and k else set()
        for i, example in enumerate(examples):
            if i in moved:
                splits["train"].append(example)
                splits["dev"].append(example)
            else:
                splits[split].append(example)
    return splits


class GroundedScanDataset:
    """seq2seq/gSCAN_dataset.py:105-310 with the same constructor, attributes and iterator tuple."""

    def __init__(self, path_to_data: str, save_directory: str, k: int, split="train", input_vocabulary_file="",
                 target_vocabulary_file="", generate_vocabulary=False, vocabularies=None):
        """`vocabularies` (additive): an (input, target) pair of Vocabulary objects to share with another split
        instead of reading them back from the files the training split has just written."""
        assert os.path.exists(path_to_data), "Trying to read a gSCAN dataset from a non-existing file {}.".format(
            path_to_data)
        if not generate_vocabulary and vocabularies is None:
            assert os.path.exists(os.path.join(save_directory, input_vocabulary_file)) and os.path.exists(
                os.path.join(save_directory, target_vocabulary_file)), \
                "Trying to load vocabularies from non-existing files."
        if split == "test" and generate_vocabulary:
            logger.warning("WARNING: generating a vocabulary from the test set.")
        self._raw = load_examples(path_to_data, k=k)[split]
        self.image_dimensions = None
        self.image_channels = 3
        self.split = split
        self.directory = save_directory
        self._order = np.zeros(0, dtype=np.int64)
        self._input_lengths = np.zeros(0, dtype=np.int64)
        self._target_lengths = np.zeros(0, dtype=np.int64)
        if vocabularies is not None:
            self.input_vocabulary, self.target_vocabulary = vocabularies
        elif generate_vocabulary:
            logger.info("Generating vocabularies...")
            self.input_vocabulary, self.target_vocabulary = Vocabulary(), Vocabulary()
            for example in self._raw:                                        # read_vocabularies, :153-160
                self.input_vocabulary.add_sentence(parse_command_repr(example["command"]))
                self.target_vocabulary.add_sentence(parse_command_repr(example["target_commands"]))
            logger.info("Done generating vocabularies.")
        else:
            logger.info("Loading vocabularies...")
            self.input_vocabulary = Vocabulary.load(os.path.join(save_directory, input_vocabulary_file))
            self.target_vocabulary = Vocabulary.load(os.path.join(save_directory, target_vocabulary_file))
            logger.info("Done loading vocabularies.")

    # ---- vocabulary plumbing (:162-176, :280-310) ------------------------------------------------
    def save_vocabularies(self, input_vocabulary_file: str, target_vocabulary_file: str):
        self.input_vocabulary.save(os.path.join(self.directory, input_vocabulary_file))
        self.target_vocabulary.save(os.path.join(self.directory, target_vocabulary_file))

    def get_vocabulary(self, vocabulary: str) -> Vocabulary:
        if vocabulary == "input":
            return self.input_vocabulary
        if vocabulary == "target":
            return self.target_vocabulary
        raise ValueError("Specified unknown vocabulary in sentence_to_array: {}".format(vocabulary))

    def sentence_to_array(self, sentence: List[str], vocabulary: str) -> List[int]:
        vocab = self.get_vocabulary(vocabulary)
        return [vocab.sos_idx] + [vocab.word_to_idx(word) for word in sentence] + [vocab.eos_idx]

    def array_to_sentence(self, sentence_array: List[int], vocabulary: str) -> List[str]:
        vocab = self.get_vocabulary(vocabulary)
        return [vocab.idx_to_word(int(idx)) for idx in sentence_array]

    num_examples = property(lambda self: len(self._order))
    input_vocabulary_size = property(lambda self: self.input_vocabulary.size)
    target_vocabulary_size = property(lambda self: self.target_vocabulary.size)

    # ---- packing (:233-278) ---------------------------------------------------------------------
    def read_dataset(self, max_examples=None, simple_situation_representation=True) -> None:
        """Convert the split to packed host arrays.  `max_examples` keeps the reference's off-by-one: reading stops
        once MORE than max_examples examples are held (gSCAN_dataset.py:243-245)."""
        if not simple_situation_representation:
            raise NotImplementedError("Full RGB input image not implemented (the reference CLI rejects it too).")
        logger.info("Converting dataset to tensors...")
        raw = self._raw if not max_examples else self._raw[:max_examples + 1]
        inputs = [self.sentence_to_array(parse_command_repr(e["command"]), "input") for e in raw]
        targets = [self.sentence_to_array(parse_command_repr(e["target_commands"]), "target") for e in raw]
        n = len(raw)
        self._input_lengths = np.array([len(x) for x in inputs], dtype=np.int64)
        self._target_lengths = np.array([len(x) for x in targets], dtype=np.int64)
        self._commands = np.zeros((n, int(self._input_lengths.max(initial=0))), dtype=np.int64)
        self._targets = np.zeros((n, int(self._target_lengths.max(initial=0))), dtype=np.int64)
        for i, (x, y) in enumerate(zip(inputs, targets)):
            self._commands[i, :len(x)] = x
            self._targets[i, :len(y)] = y
        grids = [encode_situation(e["situation"]) for e in raw]
        self._grids = np.stack(grids) if grids else np.zeros((0, 0, 0, 0), dtype=np.uint8)
        if n:
            self.image_dimensions, self.image_channels = int(self._grids.shape[1]), int(self._grids.shape[-1])
        sit = [e["situation"] for e in raw]
        self._agent_positions = np.array(
            [int(s["agent_position"]["row"]) * int(s["grid_size"]) + int(s["agent_position"]["column"]) for s in sit],
            dtype=np.int64)
        self._target_positions = np.array(
            [int(s["target_object"]["position"]["row"]) * int(s["grid_size"]) +
             int(s["target_object"]["position"]["column"]) for s in sit], dtype=np.int64)
        self._situations = sit
        self._derivations = [e.get("derivation") for e in raw]
        self._order = np.arange(n, dtype=np.int64)

    # ---- batches (:184-231) ---------------------------------------------------------------------
    def shuffle_data(self, bucket_batches: int = 0, batch_size: int = 0) -> None:
        """:177-183 (np.random.permutation, unseeded there as well).  bucket_batches > 0 (additive): after the shuffle,
        every window of bucket_batches * batch_size examples is ordered by target length, so that the batches cut
        from it hold rows of similar length and pad less (the decoder runs T = longest row steps for every row,
        seq2seq_model.py:473); windows and their contents are still random."""
        order = self._order[np.random.permutation(len(self._order))]
        if bucket_batches > 0 and batch_size > 0:
            window = bucket_batches * batch_size
            for lo in range(0, len(order), window):
                part = order[lo:lo + window]
                order[lo:lo + window] = part[np.argsort(self._target_lengths[part], kind="stable")]
            # the batches of a window in random order (otherwise lengths would rise steadily inside every window).
            # Only the FULL batches are permuted: _index_batches cuts at multiples of batch_size, so a short batch
            # anywhere but at the end would make every later batch straddle two length-sorted ones (about half of the
            # padding saved is lost again when N % batch_size != 0).  The short remainder — a random sample of the
            # last window's rows, so that no length is excluded from the full batches systematically — stays the
            # epoch's last batch, as in the reference (gSCAN_dataset.py:195-196).
            full, rest = divmod(len(order), batch_size)
            if rest:
                last = np.arange(full * batch_size - (full * batch_size) % window, len(order))   # rows of the last window
                keep = np.sort(np.random.permutation(len(last))[:len(last) - rest])              # stay sorted by length
                drop = np.setdiff1d(np.arange(len(last)), keep)
                order = np.concatenate([order[:last[0]], order[last[keep]], order[last[drop]]])
            perm = np.random.permutation(full)
            order = np.concatenate([order[b * batch_size:(b + 1) * batch_size] for b in perm] + [order[full * batch_size:]])
        self._order = order

    def _index_batches(self, batch_size: int, shard: Tuple[int, int]):
        for number, lo in enumerate(range(0, len(self._order), batch_size)):
            if number % shard[1] != shard[0]:      # shard = (rank, world): every world-th batch (evaluation under DP)
                continue
            yield self._order[lo:lo + batch_size]

    def batches(self, batch_size: int, device: Optional[torch.device] = None, shard: Tuple[int, int] = (0, 1),
                stager: Optional["BatchStager"] = None, row_shard: Tuple[int, int] = (0, 1)
                ) -> Iterator[Dict[str, torch.Tensor]]:
        """Device batches for TrainStep / greedy_decode: commands, targets, target_positions, agent_positions (int64),
        cmd_lengths, tgt_lengths (int32), world (UINT8 [B,G,G,C]) — views of one device slab that arrived in one
        asynchronous copy, a batch ahead of the consumer; plus `index` (host int64: the examples' positions in the
        split).  row_shard = (rank, world): only this rank's rows of every batch.  A batch's tensors stay valid until the iterator is advanced ONCE more (the device slab is handed back
        to the copy stream at the next delivery; keep a batch longer and it races with the copy three batches on: clone it)."""
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if device.type != "cuda":
            raise RuntimeError("batches() stages through pinned memory onto the HIP device; use get_data_iterator() "
                               "for host tensors")
        own = stager is None
        if own:
            stager = BatchStager(device, self.slab_bytes(batch_size))
        rank, world = row_shard

        def row_sets():
            for idx in self._index_batches(batch_size, shard):
                if world > 1:
                    # data-parallel training: this rank gathers and copies ONLY its rows of the global batch (rows
                    # [floor(r B / W), floor((r+1) B / W)), as train.shard_batch cuts them); a trailing batch with fewer
                    # rows than ranks is dropped on every rank
                    if len(idx) < world:
                        continue
                    idx = idx[rank * len(idx) // world:(rank + 1) * len(idx) // world]
                yield idx

        consumer = torch.cuda.current_stream(device)
        if os.environ.get("GSCAN_BATCHER_THREAD", "0") == "1" and stager.depth >= 4:
            # Round 6, built, measured and OFF by default: the host gather of batch n + 1 on a WORKER
            # thread (numpy releases the GIL in np.take) while this thread issues the step of batch n.  The worker is
            # exactly one batch ahead (queue of one): batch n + 1 is being staged while n is consumed and n - 1's slab is
            # still untouched, which is what the iterator promises with a ring of four slabs.  Measured
            # (profiles/r06_bench_with_batcher*.json): 0.916 / 0.924 of resident batches (reference order / length buckets)
            # against 0.974 / 0.918-0.969 in line — a gather of 256 short rows is ~60 us of Python bookkeeping around ~5 us
            # of copying, so the two threads take turns on the interpreter lock instead of running side by side.  (The
            # "0.815 of resident" of round 5's bucketed leg compared the file-fed run with ONE batch — the run's last, of
            # whatever length — repeated from HBM; against 64 batches of the same order it is 0.92-0.97.)
            import queue
            import threading
            ready: "queue.Queue" = queue.Queue(maxsize=1)
            stop = threading.Event()

            def put(item) -> bool:
                while not stop.is_set():
                    try:
                        ready.put(item, timeout=0.05)
                        return True
                    except queue.Full:
                        pass
                return False

            device_index = device.index if device.index is not None else torch.cuda.current_device()

            def produce():
                try:
                    torch.cuda.set_device(device_index)
                    for idx in row_sets():
                        if not put(stager.stage(self, idx, consumer)):
                            return
                    put(None)
                except BaseException as e:             # delivered to the consumer, which re-raises it
                    put(e)

            worker = threading.Thread(target=produce, name="gscan-batcher", daemon=True)
            worker.start()
            try:
                while True:
                    item = ready.get()
                    if item is None:
                        break
                    if isinstance(item, BaseException):
                        raise item
                    yield stager.deliver(item)
            finally:
                stop.set()
                worker.join(timeout=5.0)
            return
        pending = None
        for idx in row_sets():
            staged = stager.stage(self, idx, consumer)          # host gather + asynchronous copy of the NEXT batch ...
            if pending is not None:
                yield stager.deliver(pending)         # ... while the consumer works on this one
            pending = staged
        if pending is not None:
            yield stager.deliver(pending)

    def slab_bytes(self, batch_size: int) -> int:
        """Bytes of one staging slab for batches of up to batch_size rows of this split."""
        L, T = self._commands.shape[1], self._targets.shape[1]
        cell = int(np.prod(self._grids.shape[1:])) if self._grids.ndim == 4 else 0
        return _Slab.layout(batch_size, L, T, cell)[1]

    def get_data_iterator(self, batch_size=10, device: Optional[torch.device] = None,
                          shard: Tuple[int, int] = (0, 1), world_dtype: torch.dtype = torch.float32) -> Iterator[tuple]:
        """Yields (input_batch [B,L] i64, input_lengths, derivations, situation_batch [B,G,G,C], situations,
        target_batch [B,T] i64, target_lengths, agent_positions [B] i64, target_positions [B] i64), every batch
        padded to ITS longest sequences, the last one short (as the reference).  situation_batch is float32 as in the
        reference (gSCAN_dataset.py:262-264) unless world_dtype=torch.uint8 is asked for (the HIP Model takes both)."""
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if device.type == "cuda":
            # the reference yields tensors of their own (a caller may keep them): copies of the ring's views
            for b in self.batches(batch_size, device, shard):
                idx = b["index"]
                world = b["world"].clone() if world_dtype == torch.uint8 else b["world"].to(world_dtype)
                yield (b["commands"].clone(), self._input_lengths[idx], [self._derivations[i] for i in idx], world,
                       [self._situations[i] for i in idx], b["targets"].clone(), self._target_lengths[idx],
                       b["agent_positions"].clone(), b["target_positions"].clone())
            return
        for idx in self._index_batches(batch_size, shard):      # host tensors (tests without a GPU)
            in_len, tgt_len = self._input_lengths[idx], self._target_lengths[idx]
            L, T = int(in_len.max()), int(tgt_len.max())
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
            yield (t(self._commands[idx, :L]), in_len, [self._derivations[i] for i in idx],
                   t(self._grids[idx]).to(world_dtype), [self._situations[i] for i in idx], t(self._targets[idx, :T]),
                   tgt_len, t(self._agent_positions[idx]), t(self._target_positions[idx]))


class _Slab:
    """Section offsets of one staging slab: [commands i64 | targets i64 | agent i64 | target pos i64 | cmd len i32 |
    tgt len i32 | grids u8], every section 64-byte aligned."""

    @staticmethod
    def layout(B: int, L: int, T: int, cell_bytes: int):
        sizes = (("commands", 8 * B * L), ("targets", 8 * B * T), ("agent_positions", 8 * B),
                 ("target_positions", 8 * B), ("cmd_lengths", 4 * B), ("tgt_lengths", 4 * B), ("world", B * cell_bytes))
        offsets, at = {}, 0
        for name, n in sizes:
            offsets[name] = (at, n)
            at += (n + 63) // 64 * 64
        return offsets, at


class BatchStager:
    """A ring of `depth` pinned host slabs and as many device slabs, and a copy stream (depth 4 since round 6: the
    worker thread of GroundedScanDataset.batches is one batch further ahead than the in-line form was).

    stage(): gathers the rows of a batch from the packed arrays straight into the next pinned slab (numpy writes
    into the pinned memory: no intermediate tensors, no per-array pin_memory() allocations) and enqueues ONE
    host-to-device copy of the used part of the slab on the copy stream.  deliver(): makes the consumer's stream
    wait for that copy and returns views of the device slab.  A device slab is reused `depth` batches later: the copy
    that overwrites it waits for an event recorded on the consumer's stream at THAT moment, i.e. for everything the
    consumer has enqueued by then — a batch's views stay valid until the iterator has been advanced `depth - 1` times
    more, whatever the consumer launched on them in between."""

    def __init__(self, device: torch.device, slab_bytes: int, depth: int = 4):
        self.device, self.depth, self.slab_bytes = device, depth, slab_bytes
        self.host = [torch.empty(slab_bytes, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.host_np = [h.numpy() for h in self.host]
        self.dev = [torch.empty(slab_bytes, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device)
        self.copied = [torch.cuda.Event() for _ in range(depth)]
        self.released = [None] * depth
        self.count = 0
        self._last = None
        self._rows = None                      # scratch for full-width row gathers (stage)
        self._views = {}              # (slot, B, L, T, grid shape) -> (used bytes, host numpy views, device tensor views)

    _SECTIONS = (("commands", np.int64, torch.int64), ("targets", np.int64, torch.int64),
                 ("agent_positions", np.int64, torch.int64), ("target_positions", np.int64, torch.int64),
                 ("cmd_lengths", np.int32, torch.int32), ("tgt_lengths", np.int32, torch.int32),
                 ("world", np.uint8, torch.uint8))

    def _layout(self, slot: int, B: int, L: int, T: int, grid_shape: tuple):
        """Views of slab `slot` for a batch shape, built once per shape (slicing / viewing tensors costs more host time
        than gathering the rows)."""
        key = (slot, B, L, T, grid_shape)
        hit = self._views.get(key)
        if hit is None:
            offsets, used = _Slab.layout(B, L, T, int(np.prod(grid_shape)))
            if used > self.slab_bytes:
                raise ValueError(f"batch of {B} rows needs {used} bytes, the staging slabs hold {self.slab_bytes}")
            shapes = {"commands": (B, L), "targets": (B, T), "agent_positions": (B,), "target_positions": (B,),
                      "cmd_lengths": (B,), "tgt_lengths": (B,), "world": (B,) + grid_shape}
            host, dev = {}, {}
            for name, np_type, torch_type in self._SECTIONS:
                off, n = offsets[name]
                host[name] = self.host_np[slot][off:off + n].view(np_type).reshape(shapes[name])
                dev[name] = self.dev[slot][off:off + n].view(torch_type).view(shapes[name])
            hit = self._views[key] = (used, host, dev, self.host[slot][:used], self.dev[slot][:used])
        return hit

    def stage(self, data: "GroundedScanDataset", idx: np.ndarray, consumer_stream=None) -> tuple:
        """consumer_stream: the stream the batches are consumed on (default: the calling thread's current stream; the
        batcher's worker thread passes the training thread's)."""
        slot = self.count % self.depth
        self.count += 1
        in_len, tgt_len = data._input_lengths[idx], data._target_lengths[idx]
        L, T = int(in_len.max()), int(tgt_len.max())
        used, host, dev, host_used, dev_used = self._layout(slot, len(idx), L, T, tuple(data._grids.shape[1:]))
        if self.count > self.depth:
            self.copied[slot].synchronize()            # the slab's previous copy has long left the host buffer
        # rows are gathered at the arrays' full width (np.take on a column-sliced, hence non-contiguous, source first
        # copies the whole source: 0.25 ms per batch once a length-bucketed batch is narrower than the split's longest
        # row), then the used columns go to the slab
        n = len(idx)
        if self._rows is None or self._rows[0].shape[0] < n or self._rows[0].shape[1] != data._commands.shape[1] \
                or self._rows[1].shape[1] != data._targets.shape[1]:
            self._rows = (np.empty((n, data._commands.shape[1]), dtype=data._commands.dtype),
                          np.empty((n, data._targets.shape[1]), dtype=data._targets.dtype))
        rows_c, rows_t = self._rows[0][:n], self._rows[1][:n]
        np.take(data._commands, idx, axis=0, out=rows_c, mode="clip")        # mode: no buffering of out
        np.take(data._targets, idx, axis=0, out=rows_t, mode="clip")
        host["commands"][...] = rows_c[:, :L]
        host["targets"][...] = rows_t[:, :T]
        np.take(data._agent_positions, idx, out=host["agent_positions"], mode="clip")
        np.take(data._target_positions, idx, out=host["target_positions"], mode="clip")
        host["cmd_lengths"][:] = in_len
        host["tgt_lengths"][:] = tgt_len
        np.take(data._grids, idx, axis=0, out=host["world"], mode="clip")
        copy = self.copy_stream
        if self.count > self.depth:                    # the slab held a batch before: everything the consumer has
            if self.released[slot] is None:            # enqueued so far (all of it older than this copy) goes first
                self.released[slot] = torch.cuda.Event()
            self.released[slot].record(consumer_stream if consumer_stream is not None else torch.cuda.current_stream(self.device))
            copy.wait_event(self.released[slot])
        with torch.cuda.stream(copy):
            dev_used.copy_(host_used, non_blocking=True)
        self.copied[slot].record(copy)
        return slot, dev, idx

    def deliver(self, staged: tuple) -> Dict[str, torch.Tensor]:
        slot, dev, idx = staged
        current = torch.cuda.current_stream(self.device)
        self._last = slot
        current.wait_event(self.copied[slot])
        return dict(dev, index=idx)
