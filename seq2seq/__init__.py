"""Drop-in package name of the reference (`from seq2seq.model import Model`, `python -m seq2seq`).
Everything resolves to the MI355X implementation in `multimodal_seq2seq_gscan_amd`."""
