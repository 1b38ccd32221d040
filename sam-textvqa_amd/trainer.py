"""Train-step harness with the reference's step semantics (train.py:133-144, sam/task_utils.py:19-57):
forward -> masked BCE -> backward -> clip_grad_norm_(0.25) -> Adam(lr groups) -> LambdaLR warm-up/decay,
running on the flat parameter storage so that clip + Adam + bf16 refresh are two kernels, and the
data-parallel gradient exchange is a few large RCCL all-reduces (parallel.py)."""
import os
from bisect import bisect

import torch

from . import ops, parallel
from .autograd import BceLossFn, DeferredWgrads, dropout_clock
from .params import prepare


def lr_lambda(it, warmup_iters=1000, warmup_factor=0.2, lr_decay_iters=(14000, 19000), lr_decay=0.1):
    """sam/task_utils.py:48-54"""
    if it <= warmup_iters:
        alpha = float(it) / float(warmup_iters)
        return warmup_factor * (1.0 - alpha) + alpha
    return pow(lr_decay, bisect(list(lr_decay_iters), it))


def masked_bce_loss(batch_dict, grad_scale=1.0, unit_grad=False, global_count=None):
    """M4CDecodingBCEWithMaskLoss on the score blocks SAM4C.forward left in batch_dict.  unit_grad=True: the caller promises to call
    .backward() on the returned loss with the default gradient of 1 (the loss gradient is then handed on without being rescaled).
    global_count: device scalar, the all-reduced number of unmasked decoding steps of the global batch (data parallel, see Trainer.step)"""
    return BceLossFn.apply(batch_dict["fixed_scores"], batch_dict["dynamic_ocr_scores"], batch_dict["targets"], batch_dict["train_loss_mask"], grad_scale,
                           unit_grad, global_count)


class Trainer:
    def __init__(self, model, base_lr=1e-4, max_grad_norm=0.25, betas=(0.9, 0.999), eps=1e-8, schedule=None, reducer=None, seed=0, use_graph=None,
                 pipeline_update=None, overlap=True):
        self.model = model
        self.base_lr = base_lr
        groups = model.get_optimizer_parameters(base_lr)
        self.group_lr = [g.get("lr", base_lr) for g in groups]
        self.flat = prepare(model, groups=[g["params"] for g in groups])
        if len(self.flat.segment_ends) != len(groups):
            raise RuntimeError("flat storage was prepared without optimizer groups; build the Trainer before the first forward")
        dev = self.flat.flat.device
        self.exp_avg = torch.zeros_like(self.flat.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat.flat)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.max_grad_norm, self.betas, self.eps = max_grad_norm, betas, eps
        self.schedule = schedule or {}
        dist = parallel.dist
        if reducer is None and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("SAM_FORCE_DIST") == "1"):
            reducer = parallel.GradReducer(self.flat.grad, sparse_range=self._sparse_table_range(), overlap=overlap,      # (overlap=False: buckets leave after the backward, an A/B)
                                           bucket_bytes=int(float(os.environ.get("SAM_BUCKET_MB", "64")) * (1 << 20)))
        self.reducer = reducer
        # CU head-room for the collectives (csrc/gemm_common.h: grid_cu_count).  Every hot kernel of the step is persistent with one block per CU; RCCL's channel
        # blocks need CUs of their own while the buckets leave underneath the BACKWARD pass (292 MB per step), and without a reserve the blocks they displace form
        # a second launch round.  Under a reducer that spans more than one rank the persistent grids launched between the loss and reducer.finish() give up
        # `cu_reserved` CUs (SAM_DP_CU_RESERVE, default parallel.DEFAULT_CU_RESERVE = the number init_distributed caps RCCL's channels at); the forward pass, which
        # has no collective beside it, keeps the whole chip.  An explicit SAM_CU_RESERVE is the floor for every launch of the process.  What the reserve costs and
        # buys: DESIGN.md section 6, profiles/r6_cu_reserve.txt.
        self._cu_base = ops.cu_reserve() if self.flat.flat.is_cuda else 0
        self.cu_reserved = self._cu_base
        multi = reducer is not None and (reducer.world_size > 1 or (reducer.force and "SAM_DP_CU_RESERVE" in os.environ))
        if self.flat.flat.is_cuda and multi:
            want = int(os.environ.get("SAM_DP_CU_RESERVE", parallel.DEFAULT_CU_RESERVE))
            self.cu_reserved = max(self._cu_base, want - want % 8)
        rank = dist.get_rank() if dist.is_initialized() else 0
        if reducer is not None and dist.is_initialized():
            # replicas must START identical and nothing re-synchronises them later: rank 0's masters and optimizer state go to everyone
            # (this also creates the RCCL communicator here, not inside the first timed step)
            for buf in (self.flat.flat, self.exp_avg, self.exp_avg_sq):
                reducer.broadcast(buf, src=0)
            self.flat.refresh_shadows()
        if reducer is not None:
            self._register_regions(reducer)
        # encoder layers (MMT + TextBert: two thirds of the parameters) have their gradients overwritten by their backward, not accumulated into
        # zeroed memory (autograd.EncoderLayerFn); SAM_GRAD_OVERWRITE=0 restores zero-fill + accumulate
        self._fresh_layers, self._fresh_ranges = [], []
        if os.environ.get("SAM_GRAD_OVERWRITE", "1") != "0":
            for lo, hi, trig in sorted(self._units()):
                if trig is not None and trig != "head" and hasattr(trig, "attention") and hasattr(trig, "intermediate"):
                    self._fresh_layers.append(trig)
                    self._fresh_ranges.append((lo, hi))
            merged = []
            for lo, hi in self._fresh_ranges:
                if merged and merged[-1][1] == lo:
                    merged[-1] = (merged[-1][0], hi)
                else:
                    merged.append((lo, hi))
            self._fresh_ranges = merged
        self.defer_ln = os.environ.get("SAM_LN_DEFER_FINALIZE", "1") != "0"
        self._grad_one = None
        # Row-sparse optimizer walk of the word table: only when no rank can receive a gradient for a row its own `touched` flags do not know --
        # i.e. without a reducer, or with one that exchanges exactly this table row-sparsely (its scatter sets the flags of every rank's rows).
        # Under a reducer that all-reduces the table densely (a caller-built GradReducer, bench.py --no-overlap) rows touched on OTHER ranks
        # arrive with non-zero gradients: skipping them in the norm / Adam would make the replicas diverge, and nothing would clear them.
        self.sparse = None
        if os.environ.get("SAM_SPARSE_ADAM", "1") != "0" and self._sparse_walk_is_safe(reducer):
            self.sparse = self._setup_sparse_table()
        self.global_step = 0
        self.epoch_id, self.current_val_score = 0, None
        self.use_graph = bool(use_graph) if use_graph is not None else os.environ.get("SAM_STEP_GRAPH", "0") == "1"
        self._graph, self._graph_sig, self._graph_warm, self._pipelined_graph = None, None, False, False
        # Captured steps only: clip + Adam of step k are the FIRST nodes of replay k + 1 instead of the last of replay k -- in two pieces: what the head of
        # the forward reads (word table, input encoders, TextBert, the heads) on the step's stream, the MMT's 42 M parameters on a stream of their own
        # underneath TextBert's forward (a chain of 30 small kernels that leaves the GPU idle).  Same arithmetic in the same order (update k is complete
        # before forward k + 1 reads a weight); between two steps the update is PENDING: flush_update() applies it (state_dict / checkpoints / an
        # eval-mode forward of the model call it).  SAM_PIPELINE_UPDATE=0 or pipeline_update=False: the update closes its own step.
        # Measured (round 4, c3 B=64): NOT a win -- any kernel running beside TextBert's chain slows the chain by 1.3-1.6x (385 -> 510-610 us even with the
        # background piece held to 32-128 blocks), which eats what the overlap saves (step 6.22 -> 6.33 ms).  Kept as an option, default OFF.
        self.pipeline_update = (os.environ.get("SAM_PIPELINE_UPDATE", "0") == "1") if pipeline_update is None else bool(pipeline_update)
        self._pending = False
        self._gate, self._gate_mirror, self._upd_stream = None, 0, None
        import weakref
        me = weakref.ref(self)

        def _flush_before_eval(module, args):
            t = me()
            if t is not None and t._pending and not module.training and not torch.cuda.is_current_stream_capturing():
                t.flush_update()
        if hasattr(model, "register_forward_pre_hook"):
            self._eval_hook = model.register_forward_pre_hook(_flush_before_eval)
        self.measure_comm, self._comm_events = False, []
        dropout_clock.manual_seed((int(seed) ^ (rank << 32)) & 0xFFFFFFFFFFFFFFFF)       # data-parallel replicas draw different masks

    # ---- data-parallel layout ------------------------------------------------------------------------------
    def _units(self):
        """(lo, hi, trigger) for every piece of the flat buffer whose gradient is final at a known point of the backward pass:
          MMT / TextBert encoder layers -> EncoderLayerFn.backward of that layer;  TextBert position / type / LayerNorm -> EmbedLayerNormFn.backward;
          PrevPredEmbeddings, classifier, pointer net -> "head": when the gradients of all three MMT inputs are complete (GradBarrierFn: every
          node downstream of them, incl. the single-node PrevPredFn and the two nn.Linear heads, has run its backward).
          object / OCR input encoders -> InputEncoderFn.backward;  the word-embedding table is exchanged row-sparsely."""
        model, flat = self.model, self.flat
        units = []

        def add(mod_or_params, trig):
            try:
                units.append(flat.range_of(mod_or_params) + (trig,))
            except (ValueError, AttributeError, IndexError):
                pass

        enc = getattr(getattr(model, "mmt", None), "encoder", None)
        for name in ("normal_layers", "spatial_layers", "implicit_layers"):
            for l in getattr(enc, name, []) if enc is not None else []:
                add(l, l)
        if getattr(model, "mmt", None) is not None and hasattr(model.mmt, "prev_pred_embeddings"):
            add(model.mmt.prev_pred_embeddings, "head")
        for attr in ("ocr_ptr_net", "classifier"):
            if hasattr(model, attr):
                add(getattr(model, attr), "head")
        tb = getattr(model, "text_bert", None)
        if tb is not None:
            for l in tb.encoder.layer:
                add(l, l)
            emb = tb.embeddings

            class _P:          # position / token-type / LayerNorm parameters of BertEmbeddings (everything but the word table)
                @staticmethod
                def parameters():
                    return [emb.position_embeddings.weight, emb.token_type_embeddings.weight, emb.LayerNorm.weight, emb.LayerNorm.bias]
            add(_P, emb.LayerNorm)
        # object / OCR input encoders: one autograd node each (InputEncoderFn), final when that node's backward returns; the region id hangs on
        # the first of the four modules
        for names in (("linear_ocr_feat_to_mmt_in", "ocr_feat_layer_norm", "linear_ocr_bbox_to_mmt_in", "ocr_bbox_layer_norm"),
                      ("linear_obj_feat_to_mmt_in", "obj_feat_layer_norm", "linear_obj_bbox_to_mmt_in", "obj_bbox_layer_norm")):
            mods = [getattr(model, n, None) for n in names]
            if all(m is not None for m in mods):
                class _E:
                    _mods = mods

                    @classmethod
                    def parameters(cls):
                        return [p for m in cls._mods for p in m.parameters()]
                add(_E, mods[0])
        return units

    def _register_regions(self, reducer):
        """Tell the reducer which address ranges become final at which explicit point of the backward pass, from the end of the buffer down
        (SAM4C._sam_param_rank lays the parameters of each optimizer group out in backward order): walk the units in descending address order
        while they tile the buffer (the row-sparse table may sit in between) and have a trigger; adjacent "head" units merge into one region.
        Whatever lies below the first gap leaves at finish()."""
        units = sorted(self._units(), key=lambda u: -u[0])
        if not units:
            return
        s_lo, s_hi = reducer.sparse_lo, reducer.sparse_hi
        regions, expect = [], self.flat.numel
        for lo, hi, trig in units:
            if hi != expect and not (hi == s_lo and expect == s_hi and s_hi > s_lo):
                break
            if regions and trig == "head" and regions[-1][2] == "head" and regions[-1][0] == hi:
                regions[-1] = (lo, regions[-1][1], "head")
            else:
                regions.append((lo, hi, trig))
            expect = lo
        if not regions:
            return
        ids = reducer.register_regions([(lo, hi) for lo, hi, _ in regions])
        heads = []
        for (lo, hi, trig), rid in zip(regions, ids):
            if trig == "head":
                heads.append(rid)
            else:
                trig._sam_region_id = rid
        if heads:
            reducer.set_barrier(("txt", "obj", "ocr"), heads)

    def _set_ln_defer(self, on):
        from . import torchops
        ops.LnFinalizeQueue.defer = on
        if torchops.enabled():
            torchops.ns().set_ln_defer(on)

    def _ln_flush(self):
        from . import torchops
        ops.LnFinalizeQueue.flush()
        if torchops.enabled():
            torchops.ns().ln_finalize_flush()

    def _ln_clear(self):
        from . import torchops
        ops.LnFinalizeQueue.clear()
        if torchops.enabled():
            torchops.ns().ln_finalize_clear()

    def _setup_sparse_table(self):
        """(lo, hi, row_len, touched) for the word-embedding table of TextBert (include/sam_hip.h: sam_sparse_rows): 23.4 M of the 96.6 M parameters, of
        which a step touches at most B * 20 rows.  Rows that have never received a gradient have g = exp_avg = exp_avg_sq = 0: torch.optim.Adam leaves
        them exactly where they are, so the norm and the update skip them (bit-identical; 0.7 GB less traffic per step while few rows are in use --
        the synthetic bench replays one batch, i.e. ~800 rows; real TextVQA questions use a few thousand of the 30522 word pieces).  The table's
        gradient is cleared by the optimizer row by row, not by the per-step zero-fill."""
        emb = getattr(getattr(getattr(self.model, "text_bert", None), "embeddings", None), "word_embeddings", None)
        w = getattr(emb, "weight", None)
        idx = getattr(w, "_sam_index", None)
        if idx is None or w.dim() != 2 or w.shape[1] % 4 or not w.is_cuda:
            return None
        lo = self.flat.layout[idx][0]
        hi = lo + w.numel()
        if lo % 4 or w.grad is None or w.grad.stride(0) != w.shape[1]:
            return None
        touched = torch.zeros(w.shape[0], dtype=torch.uint8, device=w.device)
        w.grad.zero_()
        w.grad._sam_touched = touched                     # ops.embedding_bwd[_sorted] flag the rows they add to
        return (lo, hi, w.shape[1], touched)

    def _sparse_walk_is_safe(self, reducer):
        if reducer is None:
            return True
        rng = self._sparse_table_range(mark=False)
        return rng is not None and reducer.sparse_hi > reducer.sparse_lo and reducer.sparse_lo <= rng[0] and rng[1] <= reducer.sparse_hi

    def _grad_keep_ranges(self):
        """[lo, hi) ranges the per-step zero-fill leaves alone: the encoder layers' gradients (overwritten by their backward) and the row-sparse table"""
        r = list(self._fresh_ranges)
        if self.sparse is not None:
            r.append((self.sparse[0], self.sparse[1]))
        r.sort()
        merged = []
        for lo, hi in r:
            if merged and merged[-1][1] >= lo:
                merged[-1] = (merged[-1][0], max(hi, merged[-1][1]))
            else:
                merged.append((lo, hi))
        return merged

    def _sparse_table_range(self, mark=True):
        """the word-embedding table's [lo, hi) in flat storage when it can be exchanged row-sparsely (parallel.GradReducer.sparse_rows):
        it must start its optimizer group's segment or the buffer, so that the dense ranges around it stay whole; else None"""
        emb = getattr(getattr(getattr(self.model, "text_bert", None), "embeddings", None), "word_embeddings", None)
        idx = getattr(getattr(emb, "weight", None), "_sam_index", None)
        if idx is None or idx + 1 >= len(self.flat.layout):
            return None
        if mark:
            emb.weight._sam_sparse_reduce = True
        return self.flat.layout[idx][0], self.flat.layout[idx + 1][0]

    def current_lrs(self):
        lam = lr_lambda(self.global_step, **self.schedule)     # LambdaLR: lr(step) = base * lambda(step), stepped after opt.step()
        return [lr * lam for lr in self.group_lr]

    def step(self, batch_dict):
        """one optimisation step; returns the (device, un-synchronised) loss tensor"""
        self._bump_shadow_epoch()
        if self.use_graph and (self.reducer is None or self._dp_capturable()):
            return self._graph_step(batch_dict)
        return self._eager_step(batch_dict)

    def _dp_capturable(self):
        """the data-parallel step is captured like the single-GPU one -- ONE step for every N -- when its collectives can be: RCCL ("nccl") all-reduce /
        all-gather calls are stream-capturable and the order in which the buckets leave is fixed once the regions are registered, so the captured
        graph holds the forward, the backward with the bucket all-reduces forked onto the reducer's stream at their finality points, the join, clip
        and Adam.  Not capturable: gloo (CPU-mediated: the 2-rank CPU / shared-GPU tests), the reducer's self-check (host comparisons), SAM_DP_GRAPH=0.  (A reducer without overlap is
        captured too -- its collectives are enqueued on the step's own stream: the A/B of the overlap is then between two replayed graphs.)"""
        red = self.reducer
        if os.environ.get("SAM_DP_GRAPH", "1") == "0" or red.check or not parallel.dist.is_initialized():
            return False
        # ... and ONLY when the reducer enqueues them itself (rccl.py, `red.comm`): a step captured with torch.distributed calls inside holds
        # ProcessGroupNCCL Work objects whose events the group's watchdog polls from another thread -- the capture-invalidating query that ends in
        # std::terminate (rccl.py's header; round 4 lost 191 tests to it).  Without the direct transport (SAM_RCCL_DIRECT=0, a torch without
        # _comm_ptr, ranks that did not all get a communicator: GradReducer._resolve_comm agrees on it) the data-parallel step runs eagerly.
        if red.comm is None:
            if not getattr(self, "_warned_no_direct", False) and self.use_graph:
                self._warned_no_direct = True
                import logging
                logging.getLogger(__name__).warning("data-parallel step NOT captured: the reducer has no direct RCCL communicator (SAM_RCCL_DIRECT=0 or "
                                                    "_comm_ptr unavailable); collectives go through torch.distributed, launches stay eager")
            return False
        try:
            return parallel.dist.get_backend(red.group) == "nccl"
        except Exception:
            return False

    def _eager_step(self, batch_dict, sched_dev=None, pipelined=False):
        """everything one step enqueues.  sched_dev: device tensor [lr per group, 1 - beta1^t, 1 - beta2^t]; given, the optimizer kernel reads the
        schedule from it (graph capture: by-value arguments would freeze at their capture-time values).  pipelined (capture only): the step OPENS
        with the previous step's update (gated, _issue_pending_update) and leaves its own to the next replay."""
        model, flat = self.model, self.flat
        if not pipelined:
            self.flush_update()                                  # (an eager step between replays: the pending update first)
        if not model.training:
            model.train()                                        # (recursing over ~160 modules costs 0.6 ms of host time: only when needed)
        for layer in self._fresh_layers:
            layer._sam_grad_fresh = True                         # EncoderLayerFn.backward overwrites these gradients: they are not zeroed
        if pipelined:
            self._issue_pending_update(batch_dict)               # ... which also clears every gradient it has used: no zero-fill
            # fresh dropout masks for this step (the counter / schedule half of sam_step_advance runs at the END of the step, into the real state)
            ops.step_advance(self._rng_state, self.GRAPH_OFFSET_STRIDE, self._scratch_step, self.group_lr, self._scratch_sched, betas=self.betas, **self.schedule)
        else:
            flat.zero_grad(self._grad_keep_ranges())
        if self.reducer is not None:
            self.reducer.begin_step()
        parallel.active_reducer = self.reducer
        c_global = wait_count = None
        if self.reducer is not None and parallel.dist.is_initialized():
            # the reference normalises the loss by the number of unmasked decoding steps of the WHOLE batch (nn.DataParallel gathers the
            # scores before the loss, task_utils.py:28-29): every rank contributes its RAW count now (the all-reduce of one float runs
            # underneath the forward pass); the loss kernel divides by max(global count, 1), so the all-reduce SUM of the per-rank gradients
            # is exactly the gradient of the global mean
            c_global = batch_dict["train_loss_mask"].to(device=flat.grad.device, dtype=torch.float32).sum().reshape(1)
            wait_count = self.reducer.reduce_scalar(c_global)
        batch_dict["_sam_want_scores"] = False                   # the loss kernel reads the classifier / pointer blocks separately
        model(batch_dict)
        if wait_count is not None:
            wait_count()
        loss = masked_bce_loss(batch_dict, 1.0, unit_grad=True, global_count=c_global)
        # the encoder layers' LayerNorm backwards leave their dgamma / dbeta / dbias partial sums in place; ONE launch reduces all of them after the
        # backward pass (26 finalize launches of ~6 us each otherwise).  Under a reducer a layer's gradients must be final when its region is
        # marked: the queue is then flushed at every mark (autograd.region_done: one batched launch per layer instead of two finalizes).
        defer_ln = self.defer_ln
        if defer_ln:
            self._set_ln_defer(True)
        DeferredWgrads.late_armed = True                        # (this method joins the late stream below, before anything reads a gradient)
        if self.cu_reserved != self._cu_base:
            ops.set_cu_reserve(self.cu_reserved)                # host-side state read at launch (and at capture): the backward's persistent grids leave CUs to the collectives
        try:
            if self._grad_one is None or self._grad_one.device != loss.device:
                self._grad_one = torch.ones((), dtype=loss.dtype, device=loss.device)
            loss.backward(self._grad_one)                       # (a resident 1.0: autograd would fill a fresh one every step)
        except BaseException:
            if self.cu_reserved != self._cu_base:
                ops.set_cu_reserve(self._cu_base)
            # whatever the LayerNorm backwards queued points into workspaces of a backward pass that no longer exists (under capture: into the
            # graph's private pool): drop it, or the next step's flush would reduce stale partial sums into dgamma / dbeta / dbias
            self._ln_clear()
            DeferredWgrads.clear()
            parallel.active_reducer = None
            raise
        finally:
            DeferredWgrads.late_armed = False
            if defer_ln:
                self._set_ln_defer(False)
        side = getattr(model, "_side_stream", None)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)      # TextBert / pointer-net backward ran there: join before the norm and the update
        DeferredWgrads.flush()                                  # (normally empty: TextBert's embedding block flushed it in its backward)
        if defer_ln:
            self._ln_flush()                                    # (before the join: the batched finalize runs beside the last weight-gradient launch, not behind it)
        DeferredWgrads.join()                                   # the MMT's last weight-gradient group ran on a stream of its own
        parallel.active_reducer = None
        if self.reducer is not None:
            timing = self.measure_comm and not torch.cuda.is_current_stream_capturing()      # (timing events cannot be recorded into a capture)
            if timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self.reducer.finish()                               # waits for the overlapped all-reduces
            if timing:
                e1.record()
                self._comm_events.append((e0, e1))
        if self.cu_reserved != self._cu_base:
            ops.set_cu_reserve(self._cu_base)                   # gradient norm, Adam and the next forward: the whole chip again
        ops.sumsq(flat.grad, self.gnorm_sq, sparse=self.sparse)     # global norm AFTER the all-reduce, as the reference clips reduced grads
        if pipelined:
            # t = ++step counter and the learning rates / bias corrections of THIS step's update, for the Adam pieces at the head of the next replay
            # (or flush_update); nothing reads the schedule between here and there
            ops.step_advance(self._scratch_rng, 0, self._step_dev, self.group_lr, sched_dev, betas=self.betas, **self.schedule)
        elif sched_dev is None:
            ops.adam_step(flat.flat, flat.grad, self.exp_avg, self.exp_avg_sq, flat.bf16, flat.segment_ends, self.current_lrs(),
                          self.global_step + 1, gnorm_sq=self.gnorm_sq, max_norm=self.max_grad_norm, betas=self.betas, eps=self.eps, sparse=self.sparse)
            self.global_step += 1
        else:
            ops.adam_step_dev(flat.flat, flat.grad, self.exp_avg, self.exp_avg_sq, flat.bf16, flat.segment_ends, sched_dev,
                              gnorm_sq=self.gnorm_sq, max_norm=self.max_grad_norm, betas=self.betas, eps=self.eps, sparse=self.sparse)
        return loss.detach()

    # ---- the update of a captured step, applied at the head of the next one -----------------------------------
    def _update_split(self):
        """first element of the piece that may run underneath the head of the forward: the MMT's parameters when they close the buffer"""
        try:
            lo, hi = self.flat.range_of(self.model.mmt)
        except (ValueError, AttributeError, IndexError):
            return self.flat.numel
        sp = self.sparse
        if hi != self.flat.numel or lo % 4 or lo == 0 or (sp is not None and sp[0] < lo < sp[1]):
            return self.flat.numel
        return lo

    def _adam_piece(self, lo, hi, gate, max_blocks=0):
        flat = self.flat
        ops.adam_step_range(flat.flat, flat.grad, self.exp_avg, self.exp_avg_sq, flat.bf16, flat.segment_ends, self._sched_dev, lo, hi, gnorm_sq=self.gnorm_sq,
                            max_norm=self.max_grad_norm, betas=self.betas, eps=self.eps, sparse=self.sparse, zero_grad=True, gate=gate, max_blocks=max_blocks)

    def _issue_pending_update(self, batch_dict):
        split, n = self._update_split(), self.flat.numel
        self._adam_piece(0, split, self._gate)
        if split < n:
            main = torch.cuda.current_stream()
            if self._upd_stream is None:
                self._upd_stream = torch.cuda.Stream()
            self._upd_stream.wait_stream(main)                   # after the first piece: that one has the whole memory system while the forward waits for it
            with torch.cuda.stream(self._upd_stream):
                self._adam_piece(split, n, self._gate, max_blocks=int(os.environ.get("SAM_UPDATE_BG_BLOCKS", "512")))     # (2 blocks per CU)
            ev = torch.cuda.Event()
            ev.record(self._upd_stream)
            batch_dict["_sam_upd_event"] = ev                    # MMT.forward waits for it before it reads its first MMT parameter

    def flush_update(self):
        """apply the update a captured step left pending (no-op otherwise): after it the parameters, their bf16 shadows and the optimizer state are those
        of `global_step` completed steps"""
        if not self._pending:
            return
        self._adam_piece(0, self.flat.numel, None)
        self._pending = False
        self._bump_shadow_epoch()

    # ---- the step as ONE hipGraph ---------------------------------------------------------------------------
    # ~280 launches per step cost the host 4.7 ms (Python modules, autograd nodes, dispatcher) however they are issued; the step's shapes are
    # static, so it is captured once and replayed: the host then spends ~0.2 ms per step (input copies + one graph launch).  What varies from
    # step to step lives in device memory: the dropout state (sam_set_rng_state: captured launches add a per-replay offset base to their
    # by-value offsets; the graph's first node advances it) and the optimizer schedule (sam_adam_step_dev).
    GRAPH_OFFSET_STRIDE = 1 << 20          # dropout offsets consumed per step (far above the ~40 sites of a step)

    def _flatten(self, bd):
        out = []
        for k in sorted(bd):
            v = bd[k]
            if torch.is_tensor(v):
                out.append((k, None, v))
            elif isinstance(v, dict):
                out.extend((k, kk, vv) for kk, vv in sorted(v.items()) if torch.is_tensor(vv))
        return out

    def _bump_shadow_epoch(self):
        self.flat.shadow_epoch = getattr(self.flat, "shadow_epoch", 0) + 1      # Adam rewrote the bf16 shadows

    def _graph_step(self, batch_dict):
        dev = self.flat.flat.device
        items = self._flatten(batch_dict)
        sig = tuple((k, kk, tuple(v.shape), v.dtype) for k, kk, v in items)
        if self._graph is not None and sig != self._graph_sig:
            return self._eager_step(batch_dict)                  # another shape (last partial batch): eager, the graph stays valid for the usual one
        if self._graph is None:
            if not self._graph_warm:                             # first call: a normal eager step (lazy kernel attributes, workspaces, RCCL-free init)
                self._graph_warm = True                          # ... on the stream the capture will use: the per-stream workspace caches (exchange
                self._cap_stream = torch.cuda.Stream()           # buffer of the grouped wgrad, LayerNorm scratch) are then filled before the capture
                self._cap_stream.wait_stream(torch.cuda.current_stream())     # and their zero-fills do not become graph nodes
                with torch.cuda.stream(self._cap_stream):
                    loss = self._eager_step(batch_dict)
                torch.cuda.current_stream().wait_stream(self._cap_stream)
                return loss
            err = None
            try:
                self._capture(items, sig, dev)
            except Exception as e:                               # capture is an optimisation: never lose the run over it
                err = e
            # data parallel: the step is replayed on ALL ranks or on none (a rank that replays while another enqueues the collectives one by one would
            # hang the job): the ranks agree, through the process group, after the capture has closed on every one of them
            everyone = parallel.agree(err is None, self.reducer.group) if self.reducer is not None else err is None
            if not everyone:
                import logging
                logging.getLogger(__name__).warning("hipGraph capture of the training step failed (%s); continuing eagerly",
                                                    "%s: %s" % (type(err).__name__, err) if err is not None else "on another rank")
                ops.set_rng_state(None)
                self._ln_clear()                                 # nothing a half-recorded backward queued may survive into the eager step
                ops.reset_workspaces()
                self.use_graph, self._graph = False, None
                return self._eager_step(batch_dict)
        stale = [(dst, v) for (k, kk, v), dst in zip(items, self._static_in) if v.data_ptr() != dst.data_ptr()]
        if stale:                                                   # one multi-tensor copy per dtype instead of ~20 launches (a batch that
            try:                                                    # already lives in input_buffers() needs none)
                torch._foreach_copy_([d for d, _ in stale], [v if v.device == d.device else v.to(d.device, non_blocking=True) for d, v in stale])
            except (RuntimeError, AttributeError):
                for d, v in stale:
                    d.copy_(v, non_blocking=True)
        # The step number, the learning rates and Adam's bias corrections live in device memory and are advanced by the graph's first node
        # (ops.step_advance): nothing a replay reads is written by the host, so the host may queue replays arbitrarily far ahead of the GPU.
        # Only when the host's own count was changed behind the graph's back (checkpoint resume, an eager step of another shape in between) is
        # the device counter re-seeded -- by a fill whose value travels BY VALUE in stream order.
        if self.global_step != self._dev_step_mirror:
            self._step_dev.fill_(self.global_step)
            self._dev_step_mirror = self.global_step
        if self._pipelined_graph:
            want = 1 if self._pending else 0
            if want != self._gate_mirror:                        # (steady state: the gate stays 1 and nothing is enqueued here)
                self._gate.fill_(want)
                self._gate_mirror = want
            if not self._pending:                                # a replay whose head update is gated off clears no gradient either
                self.flat.zero_grad(self._grad_keep_ranges())
        self._graph.replay()
        self._pending = self._pipelined_graph
        self.global_step += 1
        self._dev_step_mirror += 1
        return self._static_loss.clone()                         # (the static tensor is overwritten by the next replay: callers may keep what they get)

    def input_buffers(self):
        """the captured step's own input tensors as a batch_dict (None before the capture).  A data pipeline that writes the next batch straight into
        these (its host-to-device copies land here) hands `step()` tensors it recognises by address: no staging copy per input per step."""
        if self._graph is None:
            return None
        bd = {}
        for (k, kk, _), t in zip(self._graph_items, self._static_in):
            if kk is None:
                bd[k] = t
            else:
                bd.setdefault(k, {})[kk] = t
        return bd

    def _capture(self, items, sig, dev):
        static_in = [torch.empty_like(v, device=dev).copy_(v) for _, _, v in items]
        static_bd = {}
        for (k, kk, _), t in zip(items, static_in):
            if kk is None:
                static_bd[k] = t
            else:
                static_bd.setdefault(k, {})[kk] = t
        n = len(self.group_lr)
        self._sched_dev = torch.zeros(n + 2, dtype=torch.float32, device=dev)
        self._step_dev = torch.full((1,), self.global_step, dtype=torch.int64, device=dev)
        self._dev_step_mirror = self.global_step
        self._rng_state = torch.tensor([dropout_clock.seed & 0x7FFFFFFFFFFFFFFF, dropout_clock.offset + self.GRAPH_OFFSET_STRIDE], dtype=torch.int64, device=dev)
        self._pipelined_graph = self.pipeline_update
        if self._pipelined_graph:
            self.flush_update()
            self._gate = torch.zeros(1, dtype=torch.int32, device=dev)
            self._gate_mirror = 0
            self._scratch_step = torch.zeros(1, dtype=torch.int64, device=dev)           # the rng half of sam_step_advance counts / schedules into these
            self._scratch_sched = torch.zeros(n + 2, dtype=torch.float32, device=dev)
            self._scratch_rng = torch.zeros(2, dtype=torch.int64, device=dev)            # ... and the schedule half advances this
        saved_offset, dropout_clock.offset = dropout_clock.offset, 0          # by-value offsets inside the graph: 1, 2, 3, ... per site
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        parallel.quiesce_before_capture()                      # (RCCL process group alive: its watchdog must hold no outstanding Work when the capture opens)
        ops.set_rng_state(self._rng_state)
        import gc
        gc.collect()                       # (garbage that owns hipGraphs must not be collected in the middle of the capture: decoder.DecodeSession._capture)
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(g, stream=self._cap_stream, capture_error_mode=parallel.CAPTURE_ERROR_MODE):
                bd = {k: (dict(v) if isinstance(v, dict) else v) for k, v in static_bd.items()}
                if self._pipelined_graph:
                    loss = self._eager_step(bd, sched_dev=self._sched_dev, pipelined=True)
                else:
                    # first node: fresh dropout masks, the step counter, this step's learning rates and bias corrections -- all on the device
                    ops.step_advance(self._rng_state, self.GRAPH_OFFSET_STRIDE, self._step_dev, self.group_lr, self._sched_dev, betas=self.betas, **self.schedule)
                    loss = self._eager_step(bd, sched_dev=self._sched_dev)
                self._static_loss = loss
        finally:
            if gc_was_enabled:
                gc.enable()
            ops.set_rng_state(None)                                           # the captured launches keep the pointer; eager launches go back to by-value
            dropout_clock.offset = saved_offset
        self._graph, self._graph_sig, self._static_in = g, sig, static_in
        self._graph_items = [(k, kk, None) for k, kk, _ in items]

    def exposed_comm_ms(self):
        """mean GPU time of reducer.finish() (backward done -> every bucket reduced) over the steps run with measure_comm set"""
        if not self._comm_events:
            return 0.0
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._comm_events]
        self._comm_events = []
        return sum(ms) / len(ms)

    # ---- checkpoint in the reference's layout (train.py:177-187) ------------------------------------------
    def _torch_optimizer(self, with_state):
        """a torch.optim.Adam + LambdaLR over the model's parameters in the reference's group order (train.py:97-99, task_utils.py:37-57), carrying
        this trainer's state: what `optimizer.state_dict()` / `warmup_scheduler.state_dict()` of the reference would hold at this step"""
        groups = self.model.get_optimizer_parameters(self.base_lr)
        opt = torch.optim.Adam(groups, lr=self.base_lr, betas=self.betas, eps=self.eps)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda it: lr_lambda(it, **self.schedule))
        if with_state:
            for p in self.flat.params:
                i = p._sam_index
                opt.state[p] = {"step": torch.tensor(float(self.global_step)), "exp_avg": self.flat._view(self.exp_avg, i, p).clone(),
                                "exp_avg_sq": self.flat._view(self.exp_avg_sq, i, p).clone()}
            sched.last_epoch = self.global_step
            sched._step_count = self.global_step + 1
            for g, lr in zip(opt.param_groups, self.current_lrs()):
                g["lr"] = lr
            sched._last_lr = [g["lr"] for g in opt.param_groups]
        return opt, sched

    def state_dict(self, current_val_score=None, epoch_id=None):
        """the dict train.py:177-187 hands to torch.save: model_state_dict, optimizer_state_dict (torch.optim.Adam layout),
        warmup_scheduler_state_dict (LambdaLR layout), global_step, current_val_score, epoch_id"""
        self.flush_update()
        opt, sched = self._torch_optimizer(with_state=self.global_step > 0)
        return {"model_state_dict": {k: v.detach().clone().contiguous() for k, v in self.model.state_dict().items()},
                "optimizer_state_dict": opt.state_dict(),
                "warmup_scheduler_state_dict": sched.state_dict(),
                "global_step": self.global_step,
                "current_val_score": self.current_val_score if current_val_score is None else current_val_score,
                "epoch_id": self.epoch_id if epoch_id is None else epoch_id}

    def save_checkpoint(self, path, current_val_score=None, epoch_id=None):
        torch.save(self.state_dict(current_val_score, epoch_id), path)

    def load_model_state_dict(self, sd):
        """accepts an optional `module.` prefix (DataParallel checkpoints, evaluator.py:182-186)"""
        self.flush_update()
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        missing = self.model.load_state_dict(sd, strict=True)
        self.flat.refresh_shadows()
        return missing

    def load_state_dict(self, ckpt, load_optimizer=True):
        """restore from a reference-layout checkpoint dict (this package's or one written by the reference's train.py)"""
        self.load_model_state_dict(ckpt["model_state_dict"])
        self.global_step = int(ckpt.get("global_step", 0))
        self.epoch_id = int(ckpt.get("epoch_id", 0))
        self.current_val_score = ckpt.get("current_val_score")
        osd = ckpt.get("optimizer_state_dict")
        if load_optimizer and osd is not None and osd.get("state"):
            opt, _ = self._torch_optimizer(with_state=False)
            opt.load_state_dict(osd)                             # validates group sizes / shapes exactly as the reference's resume would
            self.exp_avg.zero_(); self.exp_avg_sq.zero_()
            steps = set()
            for p in self.flat.params:
                st = opt.state.get(p)
                if not st:
                    continue
                i = p._sam_index
                self.flat._view(self.exp_avg, i, p).copy_(st["exp_avg"])
                self.flat._view(self.exp_avg_sq, i, p).copy_(st["exp_avg_sq"])
                steps.add(int(st["step"]))
            if len(steps) == 1:
                self.global_step = steps.pop()                   # Adam's bias correction counts optimizer steps
            if self.sparse is not None:                          # rows that carry optimizer state keep moving under Adam: they count as touched
                lo, hi, d, touched = self.sparse
                live = (self.exp_avg[lo:hi].view(-1, d) != 0).any(1) | (self.exp_avg_sq[lo:hi].view(-1, d) != 0).any(1)
                touched.copy_(live.to(torch.uint8))
        wsd = ckpt.get("warmup_scheduler_state_dict")
        if wsd is not None and "last_epoch" in wsd and not (load_optimizer and osd):
            self.global_step = int(wsd["last_epoch"])
        return self

    def load_checkpoint(self, path, load_optimizer=True):
        return self.load_state_dict(torch.load(path, map_location="cpu", weights_only=False), load_optimizer)
