"""Greedy decoding and evaluation on the HIP path (seq2seq/predict.py:57-128, seq2seq/evaluate.py:10-24).

The reference decodes one example at a time (`get_data_iterator(batch_size=1)`); here a batch of any size is
encoded once and all rows are stepped together, each row until its own <EOS> (rows that have stopped keep
stepping on their last token and their outputs are discarded), which gives every row exactly what the reference's
per-example loop gives it.  The generator yields the reference's 8-tuple per EXAMPLE, so `evaluate` and a
`predict_and_save`-style writer consume it unchanged."""
from __future__ import annotations

import json
import logging
import time
from typing import Iterator, List, Tuple

import torch

from ._lib import GscanError

logger = logging.getLogger(__name__)


def sequence_accuracy(prediction: List[int], target: List[int]) -> float:
    """seq2seq/helpers.py:44-64: position-wise accuracy in percent, the shorter sequence padded with a mismatch."""
    n = max(len(prediction), len(target))
    if n == 0:
        return 0.0
    pred = list(prediction) + [0] * (n - len(prediction))
    tgt = list(target) + [-1] * (n - len(target))
    return 100.0 * sum(int(a == b) for a, b in zip(pred, tgt)) / n


def greedy_decode(model, commands: torch.Tensor, cmd_lengths, world: torch.Tensor, sos_idx: int, eos_idx: int,
                  max_decoding_steps: int) -> dict:
    """predict.py:82-115 for all rows of a batch: ONE library call (Model.greedy_decode: encode + the persistent
    decoder kernel with the argmax fed back in-kernel), one device-to-host copy of the results.  Returns per-row
    python lists `tokens` (the trailing <EOS> still included, as the loop produces it), `alpha_text`, `alpha_vis`
    and the summed visual attention [B, G*G]."""
    B = commands.shape[0]
    try:
        tokens, steps, alpha_text, alpha_vis, att_sum = model.greedy_decode(commands, cmd_lengths, world, sos_idx,
                                                                            eos_idx, max_decoding_steps)
    except GscanError as e:
        # The one-launch decoder keeps the composite head [V,4H] and a [V,V] table in LDS on top of the row's memories
        # and takes vocabularies up to 64 entries: a configuration the training kernels accept can exceed that.  The
        # same decoding then runs token by token through the HIP step kernel (the reference's own call sequence).
        logger.warning("one-launch greedy decoding not available for these dimensions (%s): decoding stepwise", e)
        return greedy_decode_stepwise(model, commands, cmd_lengths, world, sos_idx, eos_idx, max_decoding_steps)
    lengths = list(cmd_lengths) if not isinstance(cmd_lengths, torch.Tensor) else cmd_lengths.tolist()
    tok, n_steps, at, av = tokens.cpu(), steps.cpu().tolist(), alpha_text.cpu(), alpha_vis.cpu()
    rows = {"tokens": [], "alpha_text": [], "alpha_vis": [], "att_sum": att_sum}
    for r in range(B):
        n, L = int(n_steps[r]), int(lengths[r])
        rows["tokens"].append(tok[r, :n].tolist())
        rows["alpha_text"].append([at[r, s, :L].tolist() for s in range(n)])    # batch size 1 has no padding columns
        rows["alpha_vis"].append([av[r, s].tolist() for s in range(n)])
    return rows


def greedy_decode_stepwise(model, commands: torch.Tensor, cmd_lengths, world: torch.Tensor, sos_idx: int, eos_idx: int,
                           max_decoding_steps: int) -> dict:
    """The same decoding through the reference's own call sequence (encode_input, key_layer, initialize_hidden,
    decode_input per token: predict.py:82-112) with every row stepped until the last one has stopped: one launch and
    one host synchronisation per token.  Kept as the drop-in surface check of those methods; predict() uses
    greedy_decode above."""
    B = commands.shape[0]
    device = commands.device
    encoded = model.encode_input(commands_input=commands, commands_lengths=cmd_lengths, situations_input=world)
    keys_vis = model.visual_attention.key_layer(encoded["encoded_situations"])                  # :87-88
    keys_txt = model.textual_attention.key_layer(encoded["encoded_commands"]["encoder_outputs"])  # :89-90
    hidden = model.attention_decoder.initialize_hidden(
        model.tanh(model.enc_hidden_to_dec_hidden(encoded["hidden_states"])))                   # :95-96
    token = torch.full((B,), sos_idx, dtype=torch.long, device=device)
    active = torch.ones(B, dtype=torch.bool, device=device)
    att_sum = torch.zeros(B, keys_vis.shape[1], dtype=torch.float32, device=device)
    steps_tok, steps_at, steps_av, steps_active = [], [], [], []
    it = 0
    lengths = list(cmd_lengths) if not isinstance(cmd_lengths, torch.Tensor) else cmd_lengths.tolist()
    while it <= max_decoding_steps and bool(active.any()):                                       # :101
        output, hidden, _, alpha_text, alpha_vis = model.decode_input(
            target_token=token, hidden=hidden, encoder_outputs=keys_txt, input_lengths=lengths,
            encoded_situations=keys_vis)
        nxt = torch.log_softmax(output, dim=-1).max(dim=-1)[1]                                   # :106-107
        token = torch.where(active, nxt, token)
        att_sum += alpha_vis * active.unsqueeze(1)
        steps_tok.append(token)
        steps_at.append(alpha_text)
        steps_av.append(alpha_vis)
        steps_active.append(active)
        active = active & (nxt != eos_idx)
        it += 1
    tok = torch.stack(steps_tok, 1).cpu()
    act = torch.stack(steps_active, 1).cpu()
    at = torch.stack(steps_at, 1).cpu()
    av = torch.stack(steps_av, 1).cpu()
    rows = {"tokens": [], "alpha_text": [], "alpha_vis": [], "att_sum": att_sum}
    for r in range(B):
        n = int(act[r].sum())
        L = int(lengths[r])
        rows["tokens"].append(tok[r, :n].tolist())
        rows["alpha_text"].append([at[r, s, :L].tolist() for s in range(n)])    # batch size 1 has no padding columns
        rows["alpha_vis"].append([av[r, s].tolist() for s in range(n)])
    return rows


def predict(data_iterator: Iterator, model, max_decoding_steps: int, pad_idx: int, sos_idx: int, eos_idx: int,
            max_examples_to_evaluate=None) -> Iterator[Tuple]:
    """predict.py:57-128.  `data_iterator` yields the reference's 9-tuples (gSCAN_dataset.py:229-231) with any batch
    size; one 8-tuple is yielded per example:
    (input_sequence [1,L], derivation_spec, situation_spec, output_sequence, target_sequence [1,T],
     attention_weights_commands, attention_weights_situations, auxiliary_accuracy_target)."""
    model.eval()
    start = time.time()
    count = 0
    with torch.no_grad():
        for (input_sequence, input_lengths, derivation_spec, situation, situation_spec, target_sequence,
             target_lengths, agent_positions, target_positions) in data_iterator:
            B = input_sequence.shape[0]
            rows = greedy_decode(model, input_sequence, input_lengths, situation, sos_idx, eos_idx, max_decoding_steps)
            aux_scores = model.auxiliary_task_forward(rows["att_sum"]) if model.auxiliary_task else None
            for r in range(B):
                count += 1
                if max_examples_to_evaluate and count > max_examples_to_evaluate:
                    return
                out, at, av = rows["tokens"][r], rows["alpha_text"][r], rows["alpha_vis"][r]
                if out and out[-1] == eos_idx:                                                  # :116-119
                    out, at, av = out[:-1], at[:-1], av[:-1]
                aux_acc = 0
                if model.auxiliary_task:                                                        # :120-123
                    aux_acc = model.get_auxiliary_accuracy(aux_scores[r:r + 1], target_positions[r:r + 1])
                L, T = int(input_lengths[r]), int(target_lengths[r])
                pick = (lambda x: x[r] if isinstance(x, (list, tuple)) and len(x) == B else x)
                yield (input_sequence[r:r + 1, :L], pick(derivation_spec), pick(situation_spec), out,
                       target_sequence[r:r + 1, :T], at, av, aux_acc)
    logger.info("Predicted for {} examples.".format(count))
    logger.info("Done predicting in {} seconds.".format(time.time() - start))


def evaluate_sums(data_iterator: Iterator, model, max_decoding_steps: int, pad_idx: int, sos_idx: int, eos_idx: int,
                  max_examples_to_evaluate=None) -> Tuple[float, float, float, float]:
    """The sums behind evaluate(): (sum of token accuracies %, exact matches, sum of target-position accuracies %,
    examples) — additive over shards of a split, which is how data-parallel ranks share an evaluation."""
    acc_sum, target_sum, exact, n = 0.0, 0.0, 0, 0
    for _, _, _, output_sequence, target_sequence, _, _, aux_acc_target in predict(
            data_iterator=data_iterator, model=model, max_decoding_steps=max_decoding_steps, pad_idx=pad_idx,
            sos_idx=sos_idx, eos_idx=eos_idx, max_examples_to_evaluate=max_examples_to_evaluate):
        accuracy = sequence_accuracy(output_sequence, target_sequence[0].tolist()[1:-1])
        exact += int(accuracy == 100)
        acc_sum += accuracy
        target_sum += aux_acc_target
        n += 1
    return acc_sum, float(exact), target_sum, float(n)


def evaluate(data_iterator: Iterator, model, max_decoding_steps: int, pad_idx: int, sos_idx: int, eos_idx: int,
             max_examples_to_evaluate=None) -> Tuple[float, float, float]:
    """evaluate.py:10-24: (mean token accuracy %, exact match %, mean target-position accuracy %)."""
    acc_sum, exact, target_sum, n = evaluate_sums(data_iterator, model, max_decoding_steps, pad_idx, sos_idx, eos_idx,
                                                  max_examples_to_evaluate)
    return acc_sum / n, (exact / n) * 100, target_sum / n


def predict_and_save(dataset, model, output_file_path: str, max_decoding_steps: int, max_testing_examples=None,
                     batch_size: int = 256, **kwargs) -> str:
    """seq2seq/predict.py:16-54: decode every example of `dataset` and write the reference's record schema
    (input, prediction, derivation, target, situation, both attention traces, accuracy, exact_match,
    position_accuracy) as JSON.  Batched (the reference iterates with batch_size=1)."""
    vocab, output = dataset.target_vocabulary, []
    for (inp, derivation, situation, out_seq, tgt, aw_c, aw_s, pos_acc) in predict(
            dataset.get_data_iterator(batch_size=batch_size, world_dtype=torch.uint8), model=model,
            max_decoding_steps=max_decoding_steps,
            pad_idx=vocab.pad_idx, sos_idx=vocab.sos_idx, eos_idx=vocab.eos_idx,
            max_examples_to_evaluate=max_testing_examples):
        accuracy = sequence_accuracy(out_seq, tgt[0].tolist()[1:-1])
        output.append({"input": dataset.array_to_sentence(inp[0].tolist(), "input")[1:-1],
                       "prediction": dataset.array_to_sentence(out_seq, "target"), "derivation": derivation,
                       "target": dataset.array_to_sentence(tgt[0].tolist(), "target")[1:-1], "situation": situation,
                       "attention_weights_input": aw_c, "attention_weights_situation": aw_s, "accuracy": accuracy,
                       "exact_match": True if accuracy == 100 else False, "position_accuracy": pos_acc})
    with open(output_file_path, mode="w") as outfile:
        json.dump(output, outfile, indent=4)
    logger.info("Wrote predictions for {} examples.".format(len(output)))
    return output_file_path
