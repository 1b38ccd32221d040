"""CPU restatement of the reference's beam-search decoder (sam/beam_search.py:6-181) and of the loop that drives it
(sam/sa_m4c.py:304-314).  TEST INFRASTRUCTURE — see oracle/__init__.py.

Pinned by tests/golden/sam4c_small_c3.npz (`beam.*` arrays): tests/golden/make_golden.py runs the reference's own BeamSearch
class, this file is checked against those arrays in tests/test_oracle_golden.py.

Semantics worth stating (all visible in the reference lines cited):
  * `prev_position = indices / vocab_size` (beam_search.py:113) is INTEGER division: the reference was written for torch <= 1.4,
    where `/` on integer tensors floors (indices are non-negative, so floor == truncation).  Under torch >= 1.6 the same line yields
    a float tensor that cannot index; the golden generator restores the old operator for that one call (documented there).
  * `batch_dict["topkscores"]` starts as an INTEGER tensor of zeros ([B*k, 1], `new_full` on the int64 `train_prev_inds`,
    beam_search.py:61-65) and becomes float32 by type promotion at the first `+ value` (:126-128).
  * completed beams (sequence has EOS at position t, :140-143) are forced to emit EOS with log-probability 0 at step t (:89-93);
    at t == 0 only the first beam of every sample is live (:98-105); candidates are ranked over the flattened [beam, vocab]
    axis with `topk(sorted=True)` (:107-110); the batch's feature tensors are re-gathered by the surviving beams' source rows
    (:131-137) — a permutation inside each sample's group of identical copies;
  * the loop ends when every beam is complete or after the last decoding step; `complete_seqs = train_prev_inds[completed_ids]`
    (:145-158), where `completed_ids` is the [n, 1] result of `nonzero()` (so complete_seqs is [n, 1, S]) or, when the
    decoding steps ran out, `arange(B*k)` ([B*k, S]).
"""
import torch

BATCH_DICT_KEYS = ["pad_obj_features", "pad_obj_bboxes", "ocr_fasttext", "ocr_phoc", "pad_ocr_features", "pad_ocr_bboxes", "question_indices",
                   "question_mask", "pad_obj_mask", "pad_ocr_mask", "spatial_adj_matrices", "ocr_mmt_in", "obj_mmt_in", "question_id"]   # beam_search.py:14-29


class BeamSearch:
    def __init__(self, beam_size, bos_idx, eos_idx):
        self._decode_size = beam_size
        self._BOS_IDX, self._EOS_IDX = bos_idx, eos_idx
        self.completed_ids = None
        self.batch_dict_keys = list(BATCH_DICT_KEYS)

    # beam_search.py:31-82
    def init_batch(self, batch_dict):
        self.completed_ids = None
        k = self._decode_size
        self._batch_size = batch_dict["train_prev_inds"].shape[0]
        self._offset_mat = torch.arange(0, self._batch_size).repeat_interleave(k, dim=0) * k
        batch_dict["topkscores"] = batch_dict["train_prev_inds"].new_full((self._batch_size * k, 1), 0.0).detach()
        for key in self.batch_dict_keys + ["train_prev_inds"]:
            if key in batch_dict:
                if isinstance(batch_dict[key], dict):
                    for kk in batch_dict[key]:
                        batch_dict[key][kk] = batch_dict[key][kk].repeat_interleave(k, dim=0)
                else:
                    batch_dict[key] = batch_dict[key].repeat_interleave(k, dim=0)
        return batch_dict

    # beam_search.py:84-160
    def decode(self, batch_dict, t):
        k = self._decode_size
        vocab_size = batch_dict["scores"].shape[-1]
        current_scores = torch.log(torch.sigmoid(batch_dict["scores"][:, t, :]))
        if self.completed_ids is not None:
            current_scores[self.completed_ids, :] = -float("Inf")
            current_scores[self.completed_ids, self._EOS_IDX] = 0
        current_scores = current_scores + batch_dict["topkscores"].expand_as(current_scores)
        if t == 0:
            ignore_ids = ((torch.arange(0, self._batch_size) * k).view(-1, 1) + torch.arange(1, k).view(1, -1)).view(-1)
            current_scores[ignore_ids, :] = -float("Inf")
        value, indices = current_scores.reshape(self._batch_size, -1).topk(k, dim=-1, largest=True, sorted=True)
        prev_position = torch.div(indices, vocab_size, rounding_mode="floor").view(-1) + self._offset_mat.to(indices.device)
        new_position = (indices % vocab_size).view(-1)
        batch_dict["train_prev_inds"] = self.add_next_word(batch_dict["train_prev_inds"], prev_position, new_position, t)
        batch_dict["topkscores"] = batch_dict["topkscores"][prev_position] + value.view(-1).unsqueeze(1)
        for key in self.batch_dict_keys:
            if isinstance(batch_dict[key], dict):
                for kk in batch_dict[key]:
                    batch_dict[key][kk] = batch_dict[key][kk][prev_position]
            else:
                batch_dict[key] = batch_dict[key][prev_position]
        if t + 1 < batch_dict["train_prev_inds"].shape[1]:
            self.completed_ids = (batch_dict["train_prev_inds"][:, t + 1] == self._EOS_IDX).nonzero()
        else:
            self.completed_ids = torch.arange(batch_dict["train_prev_inds"].shape[0])
        finish = False
        if len(self.completed_ids) == self._batch_size * k or batch_dict["train_prev_inds"].shape[1] == t + 1:
            batch_dict["complete_seqs"] = batch_dict["train_prev_inds"][self.completed_ids, :]
            finish = True
        return finish, batch_dict, 0

    # beam_search.py:170-174
    @staticmethod
    def add_next_word(seqs, prev_word_inds, next_word_inds, t):
        new_seqs = seqs[prev_word_inds]
        if t + 1 < new_seqs.shape[1]:
            new_seqs[:, t + 1] = next_word_inds
        return new_seqs


def forward_beam_search(model, batch_dict, beam_size, eos_idx):
    """SAM4C.forward(batch_dict, use_beam_search=True), sa_m4c.py:179-202 + 304-314, on an oracle SAM4C in eval mode.
    NB the reference does NOT reset train_prev_inds before beam search (only the greedy loop does, sa_m4c.py:287-291): the caller's
    train_prev_inds are the decoder's first input, as upstream (its dataset feeds BOS + zeros at evaluation time)."""
    model._forward_obj_encoding(batch_dict)
    model._forward_ocr_encoding(batch_dict)
    bs = BeamSearch(beam_size, model.bos_idx, eos_idx)
    steps = batch_dict["train_prev_inds"].size(1)
    batch_dict = bs.init_batch(batch_dict)
    trace = []
    for t in range(steps):
        model._forward_mmt(batch_dict)
        model._forward_output(batch_dict)
        step_scores = batch_dict["scores"][:, t, :].clone()
        finish, batch_dict, _ = bs.decode(batch_dict, t)
        trace.append((step_scores, batch_dict["train_prev_inds"].clone(), batch_dict["topkscores"].clone()))
        if finish:
            break
    out = {"textvqa_scores": batch_dict["scores"], "complete_seqs": batch_dict["complete_seqs"].squeeze(), "topkscores": batch_dict["topkscores"].squeeze(),
           "question_id": batch_dict["question_id"].squeeze()}
    return out, batch_dict, trace
