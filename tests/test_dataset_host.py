"""The simulator-free dataset reader and batcher on CPU.

tests/golden/readme_demo_dataset.txt holds the one data example the reference publishes (README.md:78-166, the
first example of the split "situational_1") plus variations of it; the README also states what the grid tensor
of that example must contain (README.md:181-185), which is the known answer checked here."""
import json
import os
import shutil

import numpy as np
import pytest
import torch

from multimodal_seq2seq_gscan_amd.dataset import GroundedScanDataset, Vocabulary, encode_situation, load_examples

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "readme_demo_dataset.txt")


def test_grid_encoding_matches_the_readme_known_answer():
    example = json.load(open(DATA))["examples"]["situational_1"][0]
    grid = encode_situation(example["situation"])
    assert grid.shape == (4, 4, 15) and grid.dtype == np.uint8
    # README.md:181-185: channels = [size 1..4, circle, square, red, green, yellow, blue, agent, east, south, west,
    # north]; "the green square of size 4 in row 1 and column 1 ... [0,0,0,1,0,1,0,1,0,0,0,0,0,0,0]"
    assert grid[1, 1].tolist() == [0, 0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0]
    assert grid[3, 2].tolist() == [1, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0]        # red circle of size 1
    assert grid[2, 3].tolist() == [0] * 10 + [1, 1, 0, 0, 0]                            # agent, direction 0
    assert int(grid.sum()) == 4 * 3 + 2 and grid[0].sum() == 0


def test_encode_situation_known_answer_of_the_reference_test():
    """GroundedScan/dataset_test.py:666-693 (`test_encode_situation`) restated as data: the only test the reference
    holds that pins what the hot path consumes.  15x15 grid, the agent (direction 0) AND a red circle of size 2 in the
    SAME cell [7, 2], a green circle of size 4 at [3, 12].  Object vectors follow ObjectVocabulary.generate_objects
    (world.py:415-434) for the test's vocabulary (dataset_test.py:29-41: nouns circle, cylinder, square; colours red,
    blue, green, yellow; sizes 1-4): one-hot size | one-hot over shapes + colours -> 11 attributes, 16 channels.
    The dictionary is what Situation.to_representation (world.py:269-281) writes into a dataset file."""
    red_circle_2 = "01001001000"       # size 2 -> bit 1; circle -> bit 4 + 0; red -> bit 4 + 3
    green_circle_4 = "00011000010"     # size 4 -> bit 3; circle -> bit 4;     green -> bit 4 + 5
    situation = {
        "grid_size": 15, "agent_position": {"row": "7", "column": "2"}, "agent_direction": 0,
        "target_object": {"vector": red_circle_2, "position": {"row": "7", "column": "2"},
                          "object": {"shape": "circle", "color": "red", "size": "2"}},
        "distance_to_target": "0", "direction_to_target": "n",
        "placed_objects": {
            "0": {"vector": red_circle_2, "position": {"row": "7", "column": "2"},
                  "object": {"shape": "circle", "color": "red", "size": "2"}},
            "1": {"vector": green_circle_4, "position": {"row": "3", "column": "12"},
                  "object": {"shape": "circle", "color": "green", "size": "4"}}},
        "carrying_object": None}
    expected = np.zeros([15, 15, 11 + 1 + 4], dtype="uint8")           # dataset_test.py:681-688
    expected[7, 2, -5] = 1
    expected[7, 2, -4:] = np.array([1, 0, 0, 0])
    expected[7, 2, :-5] = [int(c) for c in red_circle_2]
    expected[3, 12, :-5] = [int(c) for c in green_circle_4]
    grid = encode_situation(situation)
    assert grid.dtype == np.uint8 and np.array_equal(grid, expected)
    # the counter-example: read_gscan/read_gscan.py:47,54 writes the object row OVER the agent's cell, erasing the agent
    # bit and direction; Grid.encode (minigrid.py:389-398) keeps both, and so must this reader
    overwritten = expected.copy()
    overwritten[7, 2, -5:] = 0
    assert not np.array_equal(grid, overwritten)
    assert grid[7, 2, :11].sum() == 3 and grid[7, 2, 11] == 1 and grid[7, 2, 12:].tolist() == [1, 0, 0, 0]
    # the order in which objects and agent are listed does not matter
    situation["placed_objects"] = dict(reversed(list(situation["placed_objects"].items())))
    assert np.array_equal(encode_situation(situation), expected)


def test_vocabulary_indices_and_json_round_trip(tmp_path):
    v = Vocabulary()
    v.add_sentence(["walk", "to", "a", "red", "circle"])
    v.add_sentence(["walk", "to", "a", "circle"])
    assert (v.pad_idx, v.sos_idx, v.eos_idx, v.size) == (0, 1, 2, 8)
    assert [v.word_to_idx(w) for w in ("walk", "to", "a", "red", "circle", "unseen")] == [3, 4, 5, 6, 7, 0]
    assert v.idx_to_word(6) == "red" and v.contains_word("red") and not v.contains_word("unseen")
    assert v.most_common(1) == [("walk", 2)]
    path = v.save(str(tmp_path / "vocab.txt"))
    assert set(json.load(open(path))) == {"sos_token", "eos_token", "pad_token", "idx_to_word", "word_to_idx",
                                          "word_frequencies"}                     # gSCAN_dataset.py:89-97
    w = Vocabulary.load(path)
    assert w.to_dict() == v.to_dict()


def test_reader_and_batcher_follow_the_reference_contract(tmp_path):
    work = str(tmp_path)
    with pytest.raises(AssertionError):
        GroundedScanDataset(DATA, work, k=0, split="train", input_vocabulary_file="in.txt",
                            target_vocabulary_file="out.txt", generate_vocabulary=False)   # no vocabulary files yet
    train = GroundedScanDataset(DATA, work, k=0, split="train", generate_vocabulary=True)
    train.read_dataset()
    assert train.num_examples == 8 and train.image_dimensions == 4 and train.image_channels == 15
    assert train.input_vocabulary_size == 3 + 7 and train.target_vocabulary_size == 3 + 3
    train.save_vocabularies("in.txt", "out.txt")
    dev = GroundedScanDataset(DATA, work, k=0, split="dev", input_vocabulary_file="in.txt",
                              target_vocabulary_file="out.txt")
    dev.read_dataset()
    assert dev.input_vocabulary.to_dict() == train.input_vocabulary.to_dict()
    batches = list(train.get_data_iterator(batch_size=3, device=torch.device("cpu")))
    assert [b[0].shape[0] for b in batches] == [3, 3, 2]                         # the last batch is short
    inp, in_len, deriv, world, sit, tgt, tgt_len, agent_pos, tgt_pos = batches[0]
    assert inp.dtype == torch.int64 and world.dtype == torch.float32 and world.shape == (3, 4, 4, 15)
    assert in_len.tolist() == [7, 6, 7] and inp.shape == (3, 7)                 # padded to the batch's longest row
    assert inp[0].tolist() == [1, 3, 4, 5, 6, 7, 2] and inp[1].tolist() == [1, 3, 4, 5, 7, 2, 0]
    assert tgt_len.tolist() == [7, 4, 6] and tgt[1].tolist() == [1, 3, 4, 2, 0, 0, 0]
    assert agent_pos.tolist() == [2 * 4 + 3, 0, 2 * 4 + 3] and tgt_pos.tolist() == [3 * 4 + 2, 3 * 4 + 2, 1 * 4 + 1]
    assert deriv[0].startswith("NP -> NN") and sit[1]["agent_direction"] == 3
    assert world[1, 0, 0].tolist() == [0] * 10 + [1, 0, 0, 0, 1]
    assert train.array_to_sentence(tgt[1].tolist()[:4], "target") == ["<SOS>", "turn left", "walk", "<EOS>"]
    np.random.seed(0)
    train.shuffle_data()
    again = torch.cat([b[0][:, :6] for b in train.get_data_iterator(batch_size=8, device=torch.device("cpu"))])
    assert again.shape[0] == 8
    # max_examples keeps the reference's off-by-one (reading stops once MORE than max_examples are held)
    few = GroundedScanDataset(DATA, work, k=0, split="train", generate_vocabulary=True)
    few.read_dataset(max_examples=2)
    assert few.num_examples == 3


def test_k_shot_moves_adverb_examples_into_train_and_dev():
    base = load_examples(DATA, k=0)
    assert len(base["adverb_1"]) == 3 and len(base["train"]) == 8 and len(base["dev"]) == 3
    moved = load_examples(DATA, k=2)
    assert len(moved["adverb_1"]) == 1 and len(moved["train"]) == 10 and len(moved["dev"]) == 5


def test_generated_dataset_file_round_trips_through_the_reader(tmp_path):
    """synthetic.write_dataset_file writes the reference's file format (GroundedScan/dataset.py:487-514): the reader
    packs it, the grids follow Grid.encode (one agent cell, three bits per object) and the iterator keeps the
    9-tuple contract."""
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, write_dataset_file
    path = str(tmp_path / "dataset.txt")
    write_dataset_file(path, {"train": 50, "dev": 7}, Shape(batch=1, max_command=9, max_target=12), seed=3)
    data = GroundedScanDataset(path, str(tmp_path), k=0, split="train", generate_vocabulary=True)
    data.read_dataset()
    assert data.num_examples == 50 and data.image_channels == 16 and data.image_dimensions == 6
    grids = data._grids
    assert grids.dtype == np.uint8 and grids.max() == 1
    assert (grids[..., 11].reshape(50, -1).sum(1) == 1).all()               # exactly one agent cell
    assert (grids[..., 12:].reshape(50, -1).sum(1) == 1).all()              # with exactly one direction bit
    objects = grids[..., :11].reshape(50, -1).sum(1)
    assert ((objects % 3) == 0).all() and objects.min() >= 3                # three bits per placed object
    batches = list(data.get_data_iterator(batch_size=16, device=torch.device("cpu")))
    assert [b[0].shape[0] for b in batches] == [16, 16, 16, 2]
    for b in batches:
        assert b[0].shape[1] == int(b[1].max()) and b[5].shape[1] == int(b[6].max())      # padded to ITS longest rows
        assert b[3].dtype == torch.float32 and tuple(b[3].shape[1:]) == (6, 6, 16)


def test_length_buckets_permute_the_split_and_cut_padding(tmp_path):
    """shuffle_data(bucket_batches=k): still a permutation of the split, batches hold rows of similar target length
    (fewer padded decoder steps), and bucket_batches=0 is the reference's plain shuffle."""
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, write_dataset_file
    path = str(tmp_path / "dataset.txt")
    write_dataset_file(path, {"train": 400}, Shape(batch=1, max_target=30), seed=5)
    data = GroundedScanDataset(path, str(tmp_path), k=0, split="train", generate_vocabulary=True)
    data.read_dataset()

    def padded_steps():
        return sum(int(b[6].max()) * len(b[6]) for b in data.get_data_iterator(batch_size=20, device=torch.device("cpu")))

    np.random.seed(0)
    data.shuffle_data()
    assert sorted(data._order.tolist()) == list(range(400))
    plain = padded_steps()
    data.shuffle_data(bucket_batches=5, batch_size=20)
    assert sorted(data._order.tolist()) == list(range(400))
    bucketed = padded_steps()
    live = int(data._target_lengths.sum())
    assert live <= bucketed < plain and (bucketed - live) < 0.5 * (plain - live)


def test_length_buckets_with_a_short_trailing_batch(tmp_path):
    """N % batch_size != 0: the short batch stays the LAST batch of the epoch (gSCAN_dataset.py:195-196) and every
    full batch is cut on a boundary of a length-sorted batch — the padding saved does not depend on N dividing evenly
    (a short batch shuffled into the middle made every later batch straddle two sorted ones)."""
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, write_dataset_file
    path = str(tmp_path / "dataset.txt")
    write_dataset_file(path, {"train": 413}, Shape(batch=1, max_target=30), seed=5)
    data = GroundedScanDataset(path, str(tmp_path), k=0, split="train", generate_vocabulary=True)
    data.read_dataset()
    lengths = data._target_lengths

    def spans():      # (rows, longest, shortest) per batch
        return [(len(b[6]), int(b[6].max()), int(b[6].min())) for b in data.get_data_iterator(batch_size=20, device=torch.device("cpu"))]

    np.random.seed(1)
    data.shuffle_data()
    plain = sum(n * hi for n, hi, _ in spans())
    saved = []
    for trial in range(4):
        data.shuffle_data(bucket_batches=5, batch_size=20)
        assert sorted(data._order.tolist()) == list(range(413))
        got = spans()
        assert [n for n, _, _ in got] == [20] * 20 + [13]                  # the short batch is last
        saved.append(plain - sum(n * hi for n, hi, _ in got))
        # a full batch is one fifth of a sorted 100-row window: its length span is a fraction of the split's
        spread = int(lengths.max() - lengths.min())
        assert np.mean([hi - lo for _, hi, lo in got[:20]]) < 0.45 * spread
    live = int(lengths.sum())
    assert min(saved) > 0.5 * (plain - live)


def test_staging_slab_layout():
    from multimodal_seq2seq_gscan_amd.dataset import _Slab
    offsets, total = _Slab.layout(256, 10, 20, 576)
    assert all(off % 64 == 0 for off, _ in offsets.values()) and total % 64 == 0
    assert offsets["world"][1] == 256 * 576 and offsets["commands"][1] == 256 * 10 * 8
    ends = sorted((off, off + n) for off, n in offsets.values())
    assert all(a[1] <= b[0] for a, b in zip(ends, ends[1:])) and ends[-1][1] <= total      # no overlap
    assert total < 256 * (576 + 8 * 30 + 16 + 8) + 7 * 64
