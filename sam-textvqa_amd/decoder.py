"""Auto-regressive decoding at evaluation time: the greedy loop of SAM4C._forward_mmt_and_output (sam/sa_m4c.py:285-302) and the beam search of
SAM4C._forward_beam_search + BeamSearch (sam/sa_m4c.py:304-314, sam/beam_search.py:6-181), SURVEY.md §8(f-3).

What the reference does: one FULL forward of the multimodal transformer per decoding step (12 of them), `argmax` / `topk` on the host side of
eager PyTorch in between.  Here:
  * under the prefix-LM mask (sa_m4c.py:834-844) no text / object / OCR row ever sees a decoder key, so those rows -- and their keys / values in
    every layer -- are the same in all steps: ONE full pass computes them, the steps re-run only the n_dec decoder rows of each layer against
    the cached keys / values (sam_attn_fwd_dec: the decoder rows' q|k|v stay in their own compact buffer, nothing is copied into the cache);
  * everything that does not depend on the step is computed once per batch (LayerNorm of the answer table and of the OCR rows inside
    PrevPredEmbeddings, the pointer network's keys, every mask);
  * the token selection is a kernel working in place on the decoder state (sam_greedy_pick / sam_beam_step), so a step needs no host decision;
  * the first pass and the decoding step are captured as two hipGraphs over static buffers: a batch costs one input copy, one replay of the
    first graph and n_dec - 1 replays of the second (the eager loop was bound by ~70 launches x 12 steps of host time);
  * greedy decoding goes one step further (sam_greedy_decode_steps, csrc/decode_steps.hip): step t only has ONE new row per sample -- decoder
    row t, whose token step t-1 picked and which never changes afterwards -- so steps 1..n_dec-1 run as a single persistent kernel whose phases
    (projection, attention over the cached keys / values, output projection + LayerNorm, FFN + LayerNorm, classifier, pointer scores, argmax,
    next embedding) are separated by grid barriers instead of ~56 launches per step.  The per-kernel step stays for model shapes the fused
    kernel is not built for (SAM_DECODE_FUSED=0 forces it).
Beam search keeps the reference's structure -- beam_size hypotheses per sample, candidates ranked over the flattened [beam, vocab] axis, surviving
beams re-gathered -- and its exact arithmetic, including the quirks listed in oracle/beam_search.py (integer `indices / vocab_size`, cumulative
scores that count the source beam twice, completed beams forced onto EOS).  What it does not repeat is the work the reference's layout makes
redundant (round 4: 21.8 -> 11.8 ms per batch of 64 at beam 5):
  * the beam_size copies of a sample (beam_search.py:31-82) have identical text / object / OCR rows in every layer and every step, so the batch is
    NOT expanded: the first pass runs once per sample and the steps' decoder rows read the one cached copy (sam_attn_fwd_dec_shared); the
    re-gathering of the feature tensors by `prev_position` (beam_search.py:131-137) permutes identical copies and is not executed
    (SAM_BEAM_SHARED=0: the expanded batch);
  * under the causal part of the mask the decoder rows before position t of a surviving beam are those of the beam it continues: their keys /
    values are re-gathered by source row and only row t is computed per step (DecodeSession._step_inc; scores and hidden states are collected per
    position and put into the order of the returned beams through an ancestry table); SAM_BEAM_INCREMENTAL=0 recomputes all decoder rows of
    all beams every step, as the reference does.  One difference in what nobody reads: when the search ends before the last position, the scores
    of the positions AFTER the step that ended it come from the continued decoding, not from the zero tokens the reference's last forward saw."""
import math
import os

import torch

from . import _capi as capi
from . import ops
from .autograd import BF16, _fused_qkv, _padded_views, _w
from .registry import registry

BATCH_DICT_KEYS = ["pad_obj_features", "pad_obj_bboxes", "ocr_fasttext", "ocr_phoc", "pad_ocr_features", "pad_ocr_bboxes", "question_indices", "question_mask",
                   "pad_obj_mask", "pad_ocr_mask", "spatial_adj_matrices", "ocr_mmt_in", "obj_mmt_in", "question_id"]      # sam/beam_search.py:14-29


class BeamSearch:
    """sam/beam_search.py:6-181 with the reference's method names; `decode` runs sam_beam_step on the GPU.  State lives in batch_dict exactly as
    upstream (`train_prev_inds`, `topkscores`, plus `_beam_done`: the completed beams as flags instead of an index list)."""

    def __init__(self, beam_size, eos_idx=None, bos_idx=None):
        if not 1 <= int(beam_size) <= 16:
            raise ValueError("beam_size must be in 1..16, got %r" % (beam_size,))
        self._decode_size = int(beam_size)
        self._EOS_IDX = registry.EOS_IDX if eos_idx is None else eos_idx
        self._BOS_IDX = registry.BOS_IDX if bos_idx is None else bos_idx
        if self._EOS_IDX is None:
            raise RuntimeError("BeamSearch needs registry.EOS_IDX (sam/beam_search.py:12) or an explicit eos_idx")
        self.completed_ids = None
        self.batch_dict_keys = list(BATCH_DICT_KEYS)

    def init_batch(self, batch_dict):
        """beam_search.py:31-82: every sample repeated beam_size times (interleaved), cumulative scores zero"""
        k = self._decode_size
        self.completed_ids = None
        prev = batch_dict["train_prev_inds"]
        self._batch_size = prev.shape[0]
        for key in self.batch_dict_keys + ["train_prev_inds"]:
            if key in batch_dict:
                v = batch_dict[key]
                if isinstance(v, dict):
                    batch_dict[key] = {kk: vv.repeat_interleave(k, dim=0) for kk, vv in v.items()}
                else:
                    batch_dict[key] = v.repeat_interleave(k, dim=0)
        dev = batch_dict["train_prev_inds"].device
        batch_dict["topkscores"] = torch.zeros((self._batch_size * k, 1), dtype=torch.float32, device=dev)   # (an integer zero tensor upstream until the first add)
        batch_dict["_beam_done"] = torch.zeros(self._batch_size * k, dtype=torch.uint8, device=dev)
        return batch_dict

    def decode(self, batch_dict, t):
        """beam_search.py:84-160 on batch_dict["scores"] (or the two blocks "fixed_scores" / "dynamic_ocr_scores"); returns (finish, batch_dict, 0)"""
        k, b = self._decode_size, self._batch_size
        fixed, ocr = batch_dict.get("fixed_scores"), batch_dict.get("dynamic_ocr_scores")
        prev = batch_dict["train_prev_inds"]
        s = prev.shape[1]
        if fixed is None or ocr is None or fixed.dtype != torch.float32:
            sc = batch_dict["scores"].float()
            n_ocr = batch_dict["pad_ocr_mask"].shape[1] if "pad_ocr_mask" in batch_dict else 0
            v = sc.shape[-1] - n_ocr
            fixed, ocr = sc[..., :v].reshape(b * k * s, v), sc[..., v:].reshape(b * k * s, -1)
            if ocr.shape[1] == 0:
                ocr = sc.new_zeros((b * k * s, 1)) - float("inf")
        fixed = fixed.reshape(b * k * s, -1)
        ocr = ocr.reshape(b * k * s, -1)
        if fixed.stride(1) != 1:
            fixed = fixed.contiguous()
        if ocr.stride(1) != 1:
            ocr = ocr.contiguous()
        prev = prev.contiguous()
        cum = batch_dict["topkscores"].reshape(-1).float().contiguous()
        done = batch_dict["_beam_done"]
        prev_pos = torch.empty(b * k, dtype=torch.int64, device=prev.device)
        ops.beam_step(fixed, ocr, b, k, prev, cum, done, self._EOS_IDX, t=t, prev_pos=prev_pos)
        batch_dict["train_prev_inds"], batch_dict["topkscores"], batch_dict["prev_position"] = prev, cum.view(-1, 1), prev_pos
        self.completed_ids = done.nonzero() if t + 1 < s else torch.arange(b * k, device=prev.device)      # beam_search.py:140-147
        finish = bool(len(self.completed_ids) == b * k) or s == t + 1
        if finish:
            batch_dict["complete_seqs"] = prev[self.completed_ids, :]
        return finish, batch_dict, 0

    def find_complete_inds(self, seqs, t):
        return (seqs[:, t] == self._EOS_IDX).nonzero()

    @staticmethod
    def add_next_word(seqs, prev_word_inds, next_word_inds, t):
        new_seqs = seqs[prev_word_inds]
        if t + 1 < new_seqs.shape[1]:
            new_seqs[:, t + 1] = next_word_inds
        return new_seqs


def graph_enabled():
    return os.environ.get("SAM_DECODE_GRAPH", "1") != "0"


def fused_enabled():
    return os.environ.get("SAM_DECODE_FUSED", "1") != "0"


_CU_COUNT = None


def _cu_count():
    global _CU_COUNT
    if _CU_COUNT is None:
        _CU_COUNT = int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count)
    return _CU_COUNT


_warned_fallback = False


def _log_fallback(why):
    global _warned_fallback
    if not _warned_fallback:
        _warned_fallback = True
        import logging
        logging.getLogger(__name__).warning("persistent decoding kernel unavailable (%s): decoding with the per-kernel step from here on", why)


class DecodeSession:
    """static buffers + the two captured graphs for one (model, input shapes, beam size)"""

    def __init__(self, model, batch_dict, beam=0, eos_idx=None, shared=False):
        """shared (beam search only): batch_dict holds the UNEXPANDED batch.  The beams of a sample have identical text / object / OCR rows in every
        layer and in every step (prefix-LM mask), so the first pass runs once per SAMPLE and the steps' decoder rows -- beam x more of them -- read
        the one cached copy (sam_attn_fwd_dec_shared); at t = 0 only a sample's first beam is live (beam_search.py:98-105) and all of its beams hold
        the same scores.  Same arithmetic per row as the expanded batch, a fifth of the first pass and of the cache at beam 5."""
        self.model, self.beam = model, int(beam)
        self.group = int(beam) if (shared and beam) else 1
        self.incremental = bool(beam) and os.environ.get("SAM_BEAM_INCREMENTAL", "1") != "0"
        self.eos = eos_idx
        self.items = _flatten(batch_dict)
        self.sig = signature(batch_dict, beam)
        dev = next(model.parameters()).device
        self.static = [torch.empty_like(v, device=dev) for _, _, v in self.items]
        self.bd = {}
        for (k, kk, _), t in zip(self.items, self.static):
            if kk is None:
                self.bd[k] = t
            else:
                self.bd.setdefault(k, {})[kk] = t
        self.rows_in, self.steps = self.bd["train_prev_inds"].shape
        self.rows = self.rows_in * self.group
        self.graph_first = self.graph_step = None
        self.pool = None
        self.fused = None          # decided after the first pass: (layers, desc, workspace) of sam_greedy_decode_steps, or False

    # ---- what is enqueued ------------------------------------------------------------------------------------
    def _first(self):
        """everything up to and including the first token selection"""
        m, bd = self.model, dict(self.bd)
        mmt = m.mmt
        bd["spatial_adj_matrices"] = dict(self.bd["spatial_adj_matrices"])
        bd["_sam_masks_u8"] = ops.pack_masks(bd["question_mask"], bd["pad_obj_mask"], bd["pad_ocr_mask"])
        m._forward_obj_encoding(bd)
        m._forward_ocr_encoding(bd)
        m._forward_text_bert(bd)
        self.enc = (bd["text_bert_emb"], bd["obj_mmt_in"], bd["ocr_mmt_in"])
        r, s, k = self.rows, self.steps, self.group
        r0 = self.rows_in
        prev_in = self.bd["train_prev_inds"]
        if k == 1:
            self.prev = prev_in
        else:
            if getattr(self, "_prev_k", None) is None:
                self._prev_k = torch.empty((r, s), dtype=prev_in.dtype, device=prev_in.device)     # (a buffer of the session: both graphs work on it in place)
            self.prev = self._prev_k
            self.prev.copy_(prev_in.repeat_interleave(k, dim=0))
        if self.beam == 0:                                   # sa_m4c.py:287-291: BOS, then zeros (beam search starts from the caller's tensor)
            self.prev.zero_()
            self.prev[:, 0] = m.bos_idx
        else:
            self.cum = torch.zeros(r, dtype=torch.float32, device=self.prev.device)
            self.done = torch.zeros(r, dtype=torch.uint8, device=self.prev.device)
            self.ctl = torch.zeros(4, dtype=torch.int32, device=self.prev.device)
            self.prev_pos = torch.arange(r, dtype=torch.int64, device=self.prev.device)
        # PrevPredEmbeddings: the step-invariant halves (sa_m4c.py:921-927)
        pp = mmt.prev_pred_embeddings
        ans_w = m.classifier.weight
        ans_x = ans_w.data if ans_w.is_contiguous() else ans_w.data.contiguous()
        self.n_ans = ans_x.shape[0]
        self.ans_ln = ops.layernorm_fwd(ans_x, pp.ans_layer_norm.weight, pp.ans_layer_norm.bias, pp.ans_layer_norm.variance_epsilon)[0]
        ocr_in = bd["ocr_mmt_in"]
        self.n_ocr = ocr_in.shape[1]
        ocr_x = ocr_in.reshape(r0 * self.n_ocr, -1)
        ocr_ln0 = ops.layernorm_fwd(ocr_x if ocr_x.is_contiguous() else ocr_x.contiguous(), pp.ocr_layer_norm.weight, pp.ocr_layer_norm.bias,
                                    pp.ocr_layer_norm.variance_epsilon)[0]
        # (shared beams: the small per-sample operands of the steps -- OCR rows of PrevPredEmbeddings, pointer keys, OCR mask -- are repeated per beam
        # once per batch, 25 MB each; the layers' keys / values, 45 MB per layer and sample group, are not)
        self.ocr_ln = ocr_ln0 if k == 1 else ocr_ln0.view(r0, -1).repeat_interleave(k, dim=0).view(r * self.n_ocr, -1)
        x_dec = self._dec_embed(prev_in, ocr_ln0)
        d = x_dec.shape[1]
        x = torch.cat([bd["text_bert_emb"].to(BF16), bd["obj_mmt_in"].to(BF16), bd["ocr_mmt_in"].to(BF16), x_dec.view(r0, s, d)], dim=1)
        self.n = x.shape[1]
        n_txt, n_obj = bd["question_mask"].size(-1), bd["pad_obj_mask"].size(-1)
        from .modules import AllowBits
        allow = AllowBits(ops.mask_bits_prefix_lm(bd["_sam_masks_u8"][0], s))
        self.plan = mmt.encoder._layer_plan(allow, bd, None)
        ocr_mask0 = bd["_sam_masks_u8"][2]
        seq2d, caches = mmt.encoder.infer_full(x.reshape(r0 * self.n, d).contiguous(), allow, bd, r0)
        self.caches = caches
        self.seq = seq2d.view(r0, self.n, d)
        ocr0 = n_txt + n_obj
        self.ocr0 = ocr0
        ocr_rows = self.seq[:, ocr0: ocr0 + self.n_ocr].contiguous()
        pk = m.ocr_ptr_net.key                                                  # pointer-network keys: OCR rows are step-invariant
        wv, _, bv, _, _, _ = _padded_views(pk.weight, pk.bias)
        ptr_k0 = ops.gemm(ocr_rows.view(r0 * self.n_ocr, d), wv, epilogue=capi.EPI_BIAS, bias=bv).view(r0, self.n_ocr, -1)
        y_dec = self.seq[:, self.n - s:].reshape(r0 * s, d)
        if self.incremental:
            # beam search, one NEW decoder row per step (_step_inc): per layer the decoder rows' q|k|v of the first pass, one block per beam, in two
            # buffers (re-gathered from one into the other by the surviving beams' source rows every step); scores / hidden states are collected
            # position by position and put in the order of the final beams through `anc` at the end
            d3 = caches[0][0].shape[1]
            self.dec_qkv = []
            for qkv_full, _, _ in caches:
                blk = qkv_full.view(r0, self.n, d3)[:, self.n - s:]
                blk = (blk if k == 1 else blk.repeat_interleave(k, dim=0)).reshape(r * s, d3).contiguous()
                self.dec_qkv.append((blk, torch.empty_like(blk)))
            self.ident = torch.arange(r, dtype=torch.int64, device=self.prev.device)
            self.anc = self.ident.repeat(s, 1)
            yd = self.seq[:, self.n - s:]
            self.y_hist = (yd if k == 1 else yd.repeat_interleave(k, dim=0)).contiguous()
        if k == 1:
            self.ptr_k, self.ocr_mask = ptr_k0, ocr_mask0
            self.out_first = self._head_and_pick(y_dec, 0 if self.incremental else None)
        else:
            fixed0, dyn0 = self._head(y_dec, ptr_k0, ocr_mask0, r0)
            self.ptr_k, self.ocr_mask = ptr_k0.repeat_interleave(k, dim=0), ocr_mask0.repeat_interleave(k, dim=0)
            fixed = fixed0.reshape(r0, -1).repeat_interleave(k, dim=0).view(r * s, -1)           # every beam of a sample: the sample's scores
            dyn = dyn0.repeat_interleave(k, dim=0)
            self.out_first = (fixed, dyn)
            self._pick(fixed, dyn.view(r * s, -1), 0 if self.incremental else None)
        if self.fused is None or self.fused:
            self.fused = self._fused_plan()         # (rebuilt on every _first: the capture's allocations replace the eager round's)

    def _dec_embed(self, prev=None, ocr_ln=None):
        """PrevPredEmbeddings.forward for the current train_prev_inds, eval mode (sa_m4c.py:928-948) -> bf16 [R*S, D]"""
        pp = self.model.mmt.prev_pred_embeddings
        prev = self.prev if prev is None else prev
        ocr_ln = self.ocr_ln if ocr_ln is None else ocr_ln
        r, s = prev.shape
        is_ocr = prev.ge(self.n_ans).view(torch.uint8).reshape(-1)
        e = ops.embed_sum_fwd(pp.position_embeddings.weight.data, pp.token_type_embeddings.weight.data, r * s, s, type_ids=is_ocr)
        emb = ops.layernorm_fwd(e, pp.emb_layer_norm.weight, pp.emb_layer_norm.bias, pp.emb_layer_norm.variance_epsilon)[0]
        return ops.gather2_add_fwd(self.ans_ln, ocr_ln, prev, self.n_ocr, emb, 0.0)

    def _step(self):
        """one decoding step: the decoder rows through every layer against the cached encoder keys / values, then the token selection"""
        from .modules import _layer_tail
        r, s = self.rows, self.steps
        x = self._dec_embed()
        for (layer, bits), (qkv_full, _, _) in zip(self.plan, self.caches):
            att = layer.attention.self
            wqkv, bqkv, _, _ = _fused_qkv(att)
            qkv_dec = ops.gemm(x, wqkv, epilogue=capi.EPI_BIAS, bias=bqkv)
            ctx = ops.attn_fwd_dec(qkv_full, qkv_dec, bits, r, self.n, s, att.num_attention_heads, 1.0 / math.sqrt(att.attention_head_size), kv_group=self.group)
            x = _layer_tail(layer, ctx, x)
        self.y_dec = x
        self.out_step = self._head_and_pick(x)

    def _fused_plan(self):
        """arguments of sam_greedy_decode_steps over this session's static buffers (after _first), or False when the fused kernel does not apply.
        The weights live re-tiled into the MFMA fragment layout in static buffers of the session (see _retile_if_stale)."""
        m, mmt = self.model, self.model.mmt
        r, s = self.rows, self.steps
        if self.beam or s < 2 or not fused_enabled():
            return False
        if r > 128 or _cu_count() % 8 != 0:          # the kernel's own limits: 16 rows per XCD x 8 XCDs, a CU count dealt evenly over the eight XCDs
            return False
        layers = []
        eps = None
        srcs = []                                   # the row-major bf16 weights, in the order of self._tiled
        for (layer, bits), (qkv_full, _, _) in zip(self.plan, self.caches):
            att, so, inter, out = layer.attention.self, layer.attention.output, layer.intermediate, layer.output
            wqkv, bqkv, _, _ = _fused_qkv(att)
            d3, d = wqkv.shape
            if d != 768 or inter.dense.weight.shape[0] != 3072 or att.attention_head_size != 64 or d3 != 3 * d:
                return False
            if getattr(att, "use_bias", False):         # head biases (sa_m4c.py:600-603) are not a phase of the persistent kernel: the per-kernel step adds them (_layer_tail)
                return False
            srcs += [wqkv, _w(so.dense.weight), _w(inter.dense.weight), _w(out.dense.weight)]
            e1, e2 = float(so.LayerNorm.variance_epsilon), float(out.LayerNorm.variance_epsilon)
            if eps is None:
                eps = e1
            if e1 != eps or e2 != eps:
                return False
            layers.append({"wqkv": None, "bqkv": bqkv, "wo": None, "bo": so.dense.bias.data, "ln1_g": so.LayerNorm.weight.data, "ln1_b": so.LayerNorm.bias.data,
                           "w1": None, "b1": inter.dense.bias.data, "w2": None, "b2": out.dense.bias.data,
                           "ln2_g": out.LayerNorm.weight.data, "ln2_b": out.LayerNorm.bias.data, "qkv": qkv_full, "allow": bits})
        if not 1 <= len(layers) <= 12 or self.n > 384 or not 1 <= self.n_ocr <= 128:
            return False
        pp = mmt.prev_pred_embeddings
        wc, _, bc, _, _, _ = _padded_views(m.classifier.weight, m.classifier.bias)
        pq = m.ocr_ptr_net.query
        wq, _, bq, _, _, _ = _padded_views(pq.weight, pq.bias)
        if wq.shape != (768, 768) or m.classifier.bias is None or pq.bias is None:
            return False
        srcs += [wc, wq]
        # fragment-tiled copies of the weights: static buffers of the session, refilled (outside the captured graphs) whenever the bf16 shadows
        # have changed since -- FlatParams.shadow_epoch, bumped by every optimizer step and every refresh -- and not once per batch (28 strided
        # copies, 156 us of a 6 ms batch)
        self._tile_srcs = srcs
        if getattr(self, "_tiled", None) is None:
            if torch.cuda.is_current_stream_capturing():
                return False                         # (cannot happen: the eager round of _capture comes first)
            self._tiled = [ops.tile_weight(w) for w in srcs]
            self._tiled_epoch = self._shadow_epoch()
        for k, l in enumerate(layers):
            l["wqkv"], l["wo"], l["w1"], l["w2"] = self._tiled[4 * k: 4 * k + 4]
        fixed, dyn = self.out_first
        att0 = self.plan[0][0].attention.self
        desc = {"n_layers": len(layers), "B": r, "N": self.n, "n_enc": self.n - s, "S": s, "H": att0.num_attention_heads, "D": 768, "F": 3072,
                "V": m.classifier.weight.shape[0], "No": self.n_ocr, "scale": 1.0 / math.sqrt(att0.attention_head_size), "ln_eps": eps,
                "emb_ln_eps": float(pp.emb_layer_norm.variance_epsilon), "ptr_scale": 1.0 / math.sqrt(m.ocr_ptr_net.query_key_size),
                "pos_emb": pp.position_embeddings.weight.data, "type_emb": pp.token_type_embeddings.weight.data, "emb_ln_g": pp.emb_layer_norm.weight.data,
                "emb_ln_b": pp.emb_layer_norm.bias.data, "ans_ln": self.ans_ln, "ocr_ln": self.ocr_ln, "wc": self._tiled[-2], "bc": bc, "wq": self._tiled[-1], "bq": bq, "ptr_k": self.ptr_k,
                "ocr_mask": self.ocr_mask, "prev_inds": self.prev, "fixed_scores": fixed if fixed.is_contiguous() else None, "ld_fixed": fixed.stride(0),
                "ocr_scores": dyn, "seq_out": self.seq}
        if desc["fixed_scores"] is None:            # a column slice of the padded logits block: same memory, row stride ld_fixed
            desc["fixed_scores"] = torch.as_strided(fixed, (fixed.shape[0] * fixed.stride(0),), (1,))
        if getattr(self, "_fused_ws", None) is None:
            self._fused_ws = ops.greedy_decode_ws(r, s, len(layers), self.prev.device)
        return layers, desc, self._fused_ws

    def _shadow_epoch(self):
        flat = getattr(self.model.classifier.weight, "_sam_flat", None)
        return getattr(flat, "shadow_epoch", None)

    def _retile_if_stale(self):
        """refill the tiled weight copies when the bf16 shadows changed since they were made (a training step, load_state_dict, ...)"""
        if not self.fused or getattr(self, "_tiled", None) is None:
            return
        ep = self._shadow_epoch()
        if ep is None or ep != self._tiled_epoch:
            for dst, w in zip(self._tiled, self._tile_srcs):
                n, k = w.shape
                dst[: n // 16].copy_(w[: n // 16 * 16].reshape(n // 16, 16, k // 8, 8).permute(0, 2, 1, 3))
                if n % 16:
                    tail = torch.zeros((16, k), dtype=w.dtype, device=w.device)
                    tail[: n % 16] = w[n // 16 * 16:]
                    dst[n // 16].copy_(tail.reshape(16, k // 8, 8).permute(1, 0, 2))
            self._tiled_epoch = ep

    def _steps_fused(self):
        """decoding steps 1 .. S-1, one launch (the score blocks, prev_inds and the final hidden states of out_first / seq are completed in place)"""
        layers, desc, ws = self.fused
        ops.greedy_decode_steps(layers, desc, ws, 1, self.steps)
        self.out_step = self.out_first
        self.y_dec = None

    def _head(self, y_dec, ptr_k, ocr_mask, r, s=None):
        """classifier + pointer network on the decoder rows (sa_m4c.py:270-278) -> (fixed f32 [r*S, V], dyn f32 [r, S, n_ocr])"""
        m = self.model
        s = self.steps if s is None else s
        wv, _, bv, _, n_pad, _ = _padded_views(m.classifier.weight, m.classifier.bias)
        fixed = ops.gemm(y_dec, wv, epilogue=capi.EPI_BIAS, bias=bv, out_dtype=torch.float32)
        if n_pad != m.classifier.weight.shape[0]:
            fixed = fixed[:, : m.classifier.weight.shape[0]]
        pq = m.ocr_ptr_net.query
        wq, _, bq, _, _, _ = _padded_views(pq.weight, pq.bias)
        q = ops.gemm(y_dec, wq, epilogue=capi.EPI_BIAS, bias=bq).view(r, s, -1)
        dyn = ops.ptr_scores_fwd(q, ptr_k, ocr_mask, 1.0 / math.sqrt(m.ocr_ptr_net.query_key_size))
        return fixed, dyn

    def _pick(self, fixed, dyn2, t=None):
        """the token selection, in place on the decoder state.  t (incremental beam steps): the position just scored -- rows 0..t of `anc` follow the
        surviving beams back to the rows that hold their scores, unless this step ended the search (the reference returns the scores of the beams as
        they were BEFORE the last re-gathering, sa_m4c.py:304-314 / beam_search.py:149-158)"""
        if self.beam == 0:
            ops.greedy_pick(fixed, dyn2, self.prev)
            return
        ops.beam_step(fixed, dyn2, self.rows // self.beam, self.beam, self.prev, self.cum, self.done, self.eos, ctl=self.ctl, prev_pos=self.prev_pos)
        if t is not None and t + 1 < self.steps:
            pp = torch.where(self.ctl[1] != 0, self.ident, self.prev_pos)
            self.anc[: t + 1] = self.anc[: t + 1].index_select(1, pp)

    def _step_inc(self, t):
        """decoding step t of a beam search, ONE new row per beam: under the causal part of the mask the decoder rows before position t of a beam are
        those of the beam it continues, so their keys / values are re-gathered by source row (what sam/beam_search.py:131-137 does to the whole batch)
        instead of being recomputed; row t goes through every layer against them and the cached encoder rows"""
        from .modules import _layer_tail
        r, s, k = self.rows, self.steps, self.group
        src, dst = (0, 1) if t % 2 == 1 else (1, 0)          # (step 1 reads the first-pass buffers)
        x_all = self._dec_embed()
        d = x_all.shape[1]
        x = x_all.view(r, s, d)[:, t]
        for (layer, bits), (qkv_full, _, _), bufs in zip(self.plan, self.caches, self.dec_qkv):
            att = layer.attention.self
            wqkv, bqkv, _, _ = _fused_qkv(att)
            cur, nxt = bufs[src], bufs[dst]
            d3 = cur.shape[1]
            torch.index_select(cur.view(r, s * d3), 0, self.prev_pos, out=nxt.view(r, s * d3))
            ops.gemm(x, wqkv, epilogue=capi.EPI_BIAS, bias=bqkv, out=nxt.view(r, s, d3)[:, t])
            scale = 1.0 / math.sqrt(att.attention_head_size)
            if os.environ.get("SAM_ATTN_DEC_ROW", "1") != "0":
                ctx_t = ops.attn_dec_row(qkv_full, nxt, bits, r, self.n, s, t, att.num_attention_heads, scale, kv_group=k)
            else:               # (the strip kernel over all decoder rows of every beam; row t of its output)
                ctx_t = ops.attn_fwd_dec(qkv_full, nxt, bits, r, self.n, s, att.num_attention_heads, scale, kv_group=k).view(r, s, -1)[:, t]
            x = _layer_tail(layer, ctx_t, x)
        self.y_hist[:, t] = x
        fixed_t, dyn_t = self._head(x, self.ptr_k, self.ocr_mask, r, 1)
        fixed, dyn = self.out_first
        ld = fixed.stride(0)
        torch.as_strided(fixed, (r, fixed.shape[1]), (s * ld, 1), fixed.storage_offset() + t * ld).copy_(fixed_t)
        dyn[:, t] = dyn_t[:, 0]
        self._pick(fixed, dyn.view(r * s, -1), t)
        self.out_step = self.out_first
        self.y_dec = None

    def _head_and_pick(self, y_dec, t=None):
        fixed, dyn = self._head(y_dec, self.ptr_k, self.ocr_mask, self.rows)
        self._pick(fixed, dyn.view(self.rows * self.steps, -1), t)
        return fixed, dyn

    # ---- running it ------------------------------------------------------------------------------------------
    def load_inputs(self, batch_dict, force=False):
        items = _flatten(batch_dict)
        stale = [(dst, v) for (_, _, v), dst in zip(items, self.static) if v.data_ptr() != dst.data_ptr()]
        if force and len(stale) != len(items):
            raise RuntimeError("DecodeSession: the first batch must not alias the session's static buffers")
        if stale:
            torch._foreach_copy_([d for d, _ in stale], [v if v.device == d.device else v.to(d.device, non_blocking=True) for d, v in stale])

    @torch.no_grad()
    def run(self, batch_dict):
        self.model._ready()
        self.load_inputs(batch_dict)
        if graph_enabled() and self.graph_first is None:
            self._capture()
            self.load_inputs(batch_dict, force=True)       # (the eager round inside _capture decoded in place on the static train_prev_inds)
        self._retile_if_stale()
        if not graph_enabled():
            self._first()
            last = self.out_first
            if self.fused:
                self._steps_fused()
            else:
                for t in range(1, self.steps):
                    self._step_inc(t) if self.incremental else self._step()
                    last = self.out_step
        else:
            self.graph_first.replay()
            last = self.out_first
            if self.incremental:
                for g in self.graph_steps:
                    g.replay()
            else:
                for _ in range(1 if self.fused else self.steps - 1):
                    self.graph_step.replay()
            if self.steps > 1:
                last = self.out_step
        if self.fused:
            inject = os.environ.get("SAM_DECODE_INJECT_ERR")          # (test hook: pretend the kernel reported this code)
            if inject:
                self._fused_ws[256] = int(inject)
            code = int(self._fused_ws[256].item())
            if code != 0:
                # the persistent launch could not own the device (a barrier timed out: other work held CUs) or the device does not deal workgroups to
                # its eight XCDs evenly: whatever it wrote is discarded and the batch is decoded again by the per-kernel step, which this session
                # keeps from here on (the reference's loop, sa_m4c.py:285-302, cannot fail this way)
                self._fused_ws.zero_()
                _log_fallback("the device does not deal the launch to eight XCDs evenly" if code == 2 else "a barrier timed out: the launch did not have the device to itself")
                last = self._fall_back(batch_dict)
        return self._results(batch_dict, last)

    def _fall_back(self, batch_dict):
        """drop the persistent kernel for this session and decode the current batch with the per-kernel step (eagerly; the graphs are re-captured,
        without the persistent kernel, on the next run)"""
        self.fused = False
        self.graph_first = self.graph_step = None
        self.load_inputs(batch_dict)
        self._first()
        last = self.out_first
        for _ in range(self.steps - 1):
            self._step()
            last = self.out_step
        return last

    def _capture(self):
        # Dead sessions (a model and its sessions form a reference cycle: only the cyclic collector frees them) own hipGraphs and their private
        # pools; a collection that happens to run in the middle of a capture destroys them there, which HIP answers with an error out of a
        # destructor -- the process aborts (seen as an intermittent "Fatal Python error: Aborted" in the test suite).  Collect now, and keep
        # the collector off until both captures are done.
        import gc
        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            self._capture_impl()
        finally:
            if was_enabled:
                gc.enable()

    def _capture_impl(self):
        # one eager round first: lazily-set kernel attributes, workspaces, flat-storage preparation must not happen inside a capture
        cur = torch.cuda.current_stream()
        st = torch.cuda.Stream()
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            self._first()
            if self.steps > 1 and self.incremental:
                for t in range(1, self.steps):
                    self._step_inc(t)
            elif self.steps > 1:
                if self.fused:
                    try:
                        self._steps_fused()
                    except capi.SamHipError as e:        # a shape / device the kernel declines after all: the per-kernel step takes over
                        _log_fallback(str(e))
                        self.fused = False
                        self._first()
                        self._step()
                else:
                    self._step()
        cur.wait_stream(st)
        torch.cuda.synchronize()
        from . import parallel
        parallel.quiesce_before_capture()        # (a live RCCL process group's watchdog must not poll into the capture: parallel.py)
        mode = parallel.CAPTURE_ERROR_MODE
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=st, capture_error_mode=mode):
            self._first()
        self.pool = g1.pool()
        g2 = torch.cuda.CUDAGraph()
        self.graph_steps = []
        if self.incremental:                     # one graph per position: its row offsets are by-value arguments
            for t in range(1, self.steps):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.pool, stream=st, capture_error_mode=mode):
                    self._step_inc(t)
                self.graph_steps.append(g)
        elif self.steps > 1:
            with torch.cuda.graph(g2, pool=self.pool, stream=st, capture_error_mode=mode):
                self._steps_fused() if self.fused else self._step()
        self.graph_first, self.graph_step = g1, g2

    def _results(self, batch_dict, last):
        fixed, dyn = last
        r, s = self.rows, self.steps
        if self.incremental and s > 1:
            # position j of a returned beam was scored when its ancestor stood in row anc[j, beam]
            rows, cols = self.anc.t(), torch.arange(s, device=self.anc.device)
            ld = fixed.stride(0)
            fixed3 = torch.as_strided(fixed, (r, s, fixed.shape[1]), (s * ld, ld, 1), fixed.storage_offset())
            scores = torch.cat([fixed3[rows, cols], dyn[rows, cols]], dim=-1)
            self.y_dec = self.y_hist[rows, cols].reshape(r * s, -1)
        else:
            scores = torch.cat([fixed.view(r, s, -1), dyn], dim=-1)
        batch_dict["scores"] = scores
        # the two blocks as views of the fresh concatenation, not of the session's static buffers (which the next batch overwrites in place: an evaluator
        # that collects per-batch outputs must not see them change).  text_bert_emb / obj_mmt_in / ocr_mmt_in below ARE the session's buffers: valid
        # until the next forward, as intermediate activations are
        n_fixed = fixed.view(r, s, -1).shape[-1]
        batch_dict["fixed_scores"], batch_dict["dynamic_ocr_scores"] = scores[..., :n_fixed], scores[..., n_fixed:]
        batch_dict["train_prev_inds"] = self.prev.clone()
        k = self.group
        batch_dict["text_bert_emb"], batch_dict["obj_mmt_in"], batch_dict["ocr_mmt_in"] = self.enc if k == 1 else tuple(t.repeat_interleave(k, dim=0) for t in self.enc)
        seq = self.seq.clone() if k == 1 else self.seq.repeat_interleave(k, dim=0)
        if self.steps > 1 and self.y_dec is not None:
            seq[:, self.n - s:] = self.y_dec.view(r, s, -1)
        batch_dict["mmt_seq_output"] = seq
        n_txt = self.bd["question_mask"].size(-1)
        batch_dict["mmt_txt_output"], batch_dict["mmt_ocr_output"] = seq[:, :n_txt], seq[:, self.ocr0: self.ocr0 + self.n_ocr]
        batch_dict["mmt_dec_output"] = seq[:, self.n - s:]
        if self.beam:
            batch_dict["topkscores"] = self.cum.clone().view(-1, 1)
            batch_dict["complete_seqs"] = batch_dict["train_prev_inds"]           # (every beam is in `completed_ids` when the reference's loop ends)
        return scores


def _flatten(bd):
    out = []
    for k in sorted(bd):
        v = bd[k]
        if k.startswith("_") or k in _OUTPUT_KEYS:
            continue
        if torch.is_tensor(v):
            out.append((k, None, v))
        elif isinstance(v, dict):
            out.extend((k, kk, vv) for kk, vv in sorted(v.items()) if torch.is_tensor(vv))
    return out


_OUTPUT_KEYS = {"scores", "fixed_scores", "dynamic_ocr_scores", "text_bert_emb", "obj_mmt_in", "ocr_mmt_in", "mmt_seq_output", "mmt_txt_output", "mmt_ocr_output",
                "mmt_dec_output", "topkscores", "complete_seqs", "prev_position", "targets", "train_loss_mask"}


def signature(batch_dict, beam):
    return (int(beam),) + tuple((k, kk, tuple(v.shape), v.dtype) for k, kk, v in _flatten(batch_dict))


def shared_beams_enabled():
    return os.environ.get("SAM_BEAM_SHARED", "1") != "0"


def session_for(model, batch_dict, beam=0, eos_idx=None, shared=False):
    """the model's cached DecodeSession for these input shapes (a handful of shapes per run: full batches and the last partial one)"""
    cache = model.__dict__.setdefault("_sam_decode_sessions", {})
    sig = signature(batch_dict, beam) + (eos_idx, bool(shared))
    ses = cache.get(sig)
    if ses is None:
        if len(cache) >= 4:
            cache.pop(next(iter(cache)))
        ses = cache[sig] = DecodeSession(model, batch_dict, beam, eos_idx, shared)
    return ses
