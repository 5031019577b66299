"""`python -m seq2seq <reference flags>` (seq2seq/__main__.py:170-172)."""
from multimodal_seq2seq_gscan_amd.__main__ import main, parser

if __name__ == "__main__":
    main(flags=vars(parser.parse_args()))
