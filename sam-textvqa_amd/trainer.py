"""Train-step harness with the reference's step semantics (train.py:133-144, sam/task_utils.py:19-57):
forward -> masked BCE -> backward -> clip_grad_norm_(0.25) -> Adam(lr groups) -> LambdaLR warm-up/decay,
running on the flat parameter storage so that clip + Adam + bf16 refresh are two kernels, and the
data-parallel gradient exchange is a few large RCCL all-reduces (parallel.py)."""
from bisect import bisect

import torch

from . import ops, parallel
from .autograd import BceLossFn, dropout_clock
from .params import prepare


def lr_lambda(it, warmup_iters=1000, warmup_factor=0.2, lr_decay_iters=(14000, 19000), lr_decay=0.1):
    """sam/task_utils.py:48-54"""
    if it <= warmup_iters:
        alpha = float(it) / float(warmup_iters)
        return warmup_factor * (1.0 - alpha) + alpha
    return pow(lr_decay, bisect(list(lr_decay_iters), it))


def masked_bce_loss(batch_dict, grad_scale=1.0, unit_grad=False, count_ratio=None):
    """M4CDecodingBCEWithMaskLoss on the score blocks SAM4C.forward left in batch_dict.  unit_grad=True: the caller promises to call
    .backward() on the returned loss with the default gradient of 1 (the loss gradient is then handed on without being rescaled).
    count_ratio: device scalar count_rank / count_global (data parallel, see Trainer.step): loss and gradient are scaled by it"""
    return BceLossFn.apply(batch_dict["fixed_scores"], batch_dict["dynamic_ocr_scores"], batch_dict["targets"], batch_dict["train_loss_mask"], grad_scale,
                           unit_grad, count_ratio)


class Trainer:
    def __init__(self, model, base_lr=1e-4, max_grad_norm=0.25, betas=(0.9, 0.999), eps=1e-8, schedule=None, reducer=None, seed=0):
        self.model = model
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
        if reducer is None and parallel.dist.is_initialized() and (parallel.dist.get_world_size() > 1 or __import__("os").environ.get("SAM_FORCE_DIST") == "1"):
            reducer = parallel.GradReducer(self.flat.grad, dense_lo=self._sparse_table_end())
        self.reducer = reducer
        if reducer is not None and parallel.dist.is_initialized() and dev.type == "cuda":
            parallel.dist.all_reduce(self.gnorm_sq)              # (zeros) creates the RCCL communicator here, not inside the first timed step
        if reducer is not None:
            self._register_regions(reducer)
        self.global_step = 0
        dropout_clock.manual_seed(seed)

    def _register_regions(self, reducer):
        """Tell the reducer which address ranges become final at which explicit point of the backward pass, from the end of the buffer down
        (SAM4C._sam_param_rank lays the parameters out in that order):
          MMT encoder layers   -> EncoderLayerFn.backward of each layer
          pointer net, classifier, PrevPredEmbeddings -> when the gradients of all three MMT inputs are complete (GradBarrierFn: every node
                                  downstream of them, incl. the single-node PrevPredFn and the two nn.Linear heads, has run its backward)
          TextBert layers      -> EncoderLayerFn.backward;   TextBert position / type / LayerNorm -> EmbedLayerNormFn.backward
        Everything below (object / OCR encoders) leaves at finish(); the word-embedding table is exchanged row-sparsely.
        Any piece that is missing or not where expected drops that region and everything below it (the ranges must tile up to the end)."""
        model, flat = self.model, self.flat
        enc = getattr(getattr(model, "mmt", None), "encoder", None)
        layers = [l for name in ("normal_layers", "spatial_layers", "implicit_layers") for l in getattr(enc, name, [])] if enc is not None else []
        if not layers:
            return
        ranges, owners = [flat.range_of(l) for l in layers], list(layers)
        try:
            lo_head, lo_enc = flat.range_of(model.ocr_ptr_net)[0], flat.range_of(enc)[0]
            tb_layers = list(model.text_bert.encoder.layer)
            tb_ranges = [flat.range_of(l) for l in tb_layers]
            emb = model.text_bert.embeddings
            lo_emb = flat.layout[emb.position_embeddings.weight._sam_index][0]
            ok = (lo_enc == min(r[0] for r in ranges) and tb_ranges[-1][1] == lo_head and tb_ranges[0][0] > lo_emb
                  and all(a[1] == b[0] for a, b in zip(tb_ranges[:-1], tb_ranges[1:]))
                  and {emb.position_embeddings.weight._sam_index, emb.token_type_embeddings.weight._sam_index, emb.LayerNorm.weight._sam_index,
                       emb.LayerNorm.bias._sam_index} == set(range(emb.position_embeddings.weight._sam_index, tb_layers[0].attention.self.query.weight._sam_index)))
        except (AttributeError, ValueError, IndexError):
            ok = False
        if ok:
            ranges += [(lo_head, lo_enc)] + tb_ranges + [(lo_emb, tb_ranges[0][0])]
            owners += ["head"] + tb_layers + [emb.LayerNorm]
        ids = reducer.register_regions(ranges)
        for owner, rid in zip(owners, ids):
            if owner == "head":
                reducer.set_barrier(("txt", "obj", "ocr"), rid)
            else:
                owner._sam_region_id = rid

    def _sparse_table_end(self):
        """if the word-embedding table is the first parameter of flat storage (it is for SAM4C: text_bert comes first) its gradient is
        exchanged row-sparsely (parallel.GradReducer.sparse_rows) and the dense all-reduce starts behind it; else 0"""
        emb = getattr(getattr(getattr(self.model, "text_bert", None), "embeddings", None), "word_embeddings", None)
        if emb is None or getattr(emb.weight, "_sam_index", None) != 0 or len(self.flat.layout) < 2:
            return 0
        emb.weight._sam_sparse_reduce = True
        return self.flat.layout[1][0]

    def current_lrs(self):
        lam = lr_lambda(self.global_step, **self.schedule)     # LambdaLR: lr(step) = base * lambda(step), stepped after opt.step()
        return [lr * lam for lr in self.group_lr]

    def step(self, batch_dict):
        """one optimisation step; returns the (device, un-synchronised) loss tensor"""
        model, flat = self.model, self.flat
        if not model.training:
            model.train()                                        # (recursing over ~160 modules costs 0.6 ms of host time: only when needed)
        flat.zero_grad()
        if self.reducer is not None:
            self.reducer.begin_step()
        parallel.active_reducer = self.reducer
        ratio = None
        if self.reducer is not None and parallel.dist.is_initialized():
            # the reference normalises the loss by the number of unmasked decoding steps of the WHOLE batch (nn.DataParallel gathers the
            # scores before the loss, task_utils.py:28-29): every rank contributes its count now (the all-reduce of one float runs
            # underneath the forward pass) and scales its loss gradient by count_rank / count_global, so that the all-reduce SUM of the
            # per-rank gradients is exactly the gradient of the global mean
            c_local = batch_dict["train_loss_mask"].to(device=flat.grad.device, dtype=torch.float32).sum().clamp_(min=1.0).reshape(1)
            c_global = c_local.clone()
            work = parallel.dist.all_reduce(c_global, async_op=True)
        model(batch_dict)
        if self.reducer is not None and parallel.dist.is_initialized():
            work.wait()
            ratio = c_local / c_global
        loss = masked_bce_loss(batch_dict, 1.0, unit_grad=True, count_ratio=ratio)
        loss.backward()
        parallel.active_reducer = None
        if self.reducer is not None:
            self.reducer.finish()                               # waits for the overlapped all-reduces
        ops.sumsq(flat.grad, self.gnorm_sq)                    # global norm AFTER the all-reduce, as the reference clips reduced grads
        ops.adam_step(flat.flat, flat.grad, self.exp_avg, self.exp_avg_sq, flat.bf16, flat.segment_ends, self.current_lrs(),
                      self.global_step + 1, gnorm_sq=self.gnorm_sq, max_norm=self.max_grad_norm, betas=self.betas, eps=self.eps)
        self.global_step += 1
        return loss.detach()

    # ---- checkpoint in the reference's layout (train.py:177-187) ------------------------------------------
    def state_dict(self):
        return {"model_state_dict": {k: v.detach().clone().contiguous() for k, v in self.model.state_dict().items()},
                "optimizer_state_dict": {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "step": self.global_step},
                "global_step": self.global_step}

    def load_model_state_dict(self, sd):
        """accepts an optional `module.` prefix (DataParallel checkpoints, evaluator.py:182-186)"""
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        missing = self.model.load_state_dict(sd, strict=True)
        self.flat.refresh_shadows()
        return missing
