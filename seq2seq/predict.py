"""Drop-in name for seq2seq/predict.py: greedy decoding on the HIP path."""
from multimodal_seq2seq_gscan_amd.predict import (greedy_decode, predict, predict_and_save,  # noqa: F401
                                                    sequence_accuracy)
