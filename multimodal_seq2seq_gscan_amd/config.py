"""Model hyper-parameter sets for the workloads named in BASELINE.md.

Keys are the keyword arguments of the reference's ``Model`` constructor
(seq2seq/model.py:26-32); values are the paper settings
(all_experiments.sh:5-7,25-27, README.md:177 for the demo)."""
from __future__ import annotations

from typing import Dict

_COMMON = dict(
    num_encoder_layers=1, encoder_bidirectional=True, num_decoder_layers=1,
    cnn_hidden_num_channels=50, input_padding_idx=0, target_pad_idx=0, target_eos_idx=2,
    output_directory="output", simple_situation_representation=True, attention_type="bahdanau",
    encoder_dropout_p=0.3, decoder_dropout_p=0.3, cnn_dropout_p=0.1,
)

WORKLOADS: Dict[str, dict] = {
    # README demo flags: 4x4 grid, hidden 20, embedding 5 -> 74 670 parameters
    "demo": dict(_COMMON, input_vocabulary_size=14, target_vocabulary_size=6, num_cnn_channels=15,
                 embedding_dimension=5, encoder_hidden_size=20, decoder_hidden_size=20, cnn_kernel_size=7,
                 conditional_attention=True, auxiliary_task=False),
    # compositional_splits / GECA: 6x6 grid, hidden 100, k=7 -> 440 275 parameters
    "compositional": dict(_COMMON, input_vocabulary_size=21, target_vocabulary_size=9, num_cnn_channels=16,
                          embedding_dimension=25, encoder_hidden_size=100, decoder_hidden_size=100,
                          cnn_kernel_size=7, conditional_attention=True, auxiliary_task=False),
    # target_length_split: k=13 -> 535 975 parameters
    "target_length": dict(_COMMON, input_vocabulary_size=17, target_vocabulary_size=8, num_cnn_channels=16,
                          embedding_dimension=25, encoder_hidden_size=100, decoder_hidden_size=100,
                          cnn_kernel_size=13, conditional_attention=True, auxiliary_task=False),
}

# Known answers published by the reference (README.md:264, adverb_run_1.txt:58, target_lengths_run_1.txt:79).
PARAMETER_TOTALS = {"demo": 74670, "compositional": 440275, "target_length": 535975}


def model_kwargs(workload: str, **overrides) -> dict:
    cfg = dict(WORKLOADS[workload])
    cfg.update(overrides)
    return cfg
