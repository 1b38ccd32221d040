"""Flat parameter storage for the HIP path.

All parameters of a prepared module tree live in ONE fp32 buffer (masters), with a same-layout fp32 gradient
buffer and a same-layout bf16 shadow buffer (the GEMM operands).  Consequences:
  * wgrad GEMMs and the fused LN/bias reductions accumulate straight into the gradient buffer (no autograd
    AccumulateGrad pass, no per-parameter tensors);
  * clip_grad_norm_ + Adam + bf16 refresh is one reduction + one elementwise kernel over the buffer;
  * data-parallel gradient exchange is a handful of large contiguous RCCL all-reduces (parallel.py).
The nn.Parameter objects stay (state_dict keys and shapes are the reference's); their .data/.grad become views.
query/key/value weights (and biases) of one attention block are laid out back to back so the fused QKV
projection reads them as one [3D, D] operand.  2-D weights whose row length is not a multiple of 8 get a
padded row stride (e.g. linear_ocr_feat_to_mmt_in [768,3002] -> stride 3008; bbox projections [768,4] -> 8).
"""
import torch

from . import ops

ALIGN = 8  # elements: 16 B in bf16, 32 B in fp32


def _round_up(x, a):
    return (x + a - 1) // a * a


def _ordered_params(module):
    """registration order (so every layer's parameters stay contiguous), except that the six q/k/v tensors of an
    attention block are emitted as [Wq, Wk, Wv, bq, bk, bv] so the fused QKV projection sees one [3D, D] operand"""
    seen, order = set(), []

    def add(p):
        if p is not None and id(p) not in seen:
            seen.add(id(p))
            order.append(p)

    qkv_of = {}
    for m in module.modules():
        if all(hasattr(m, n) for n in ("query", "key", "value")) and hasattr(m, "num_attention_heads"):
            six = [getattr(m, n).weight for n in ("query", "key", "value")] + [getattr(m, n).bias for n in ("query", "key", "value")]
            for p in six:
                qkv_of[id(p)] = six
    for p in module.parameters():
        for q in qkv_of.get(id(p), (p,)):
            add(q)
    rank = getattr(module, "_sam_param_rank", None)       # optional address-order hint (SAM4C: backward order, see modules.py); stable sort
    if rank is not None:
        names = {id(p): n for n, p in module.named_parameters()}
        order.sort(key=lambda p: rank(names[id(p)]))
    return order


class FlatParams:
    def __init__(self, module, device=None, groups=None):
        """groups: optional list of parameter lists (optimizer param groups); each becomes one contiguous segment"""
        device = torch.device(device or "cuda")
        order = _ordered_params(module)
        if groups is not None:
            gid = {}
            for gi, ps in enumerate(groups):
                for p in ps:
                    gid[id(p)] = gi
            missing = [p for p in order if id(p) not in gid]
            if missing:
                raise ValueError("%d parameters are in no optimizer group" % len(missing))
            order = sorted(order, key=lambda p: gid[id(p)])  # stable: keeps q/k/v adjacency inside a group
            self.group_of = [gid[id(p)] for p in order]
        else:
            self.group_of = [0] * len(order)
        self.params = order
        self.layout = []  # (offset, rows, cols, stride) per param; 2-D params also get zero rows up to a multiple of 8
        off = 0                                                        # (any out_features works as a GEMM N / K / M dimension)
        for p in order:
            if p.dim() == 2:
                rows, cols = p.shape
                stride = _round_up(cols, ALIGN)
                alloc_rows = _round_up(rows, ALIGN)
            else:
                rows, cols = 1, p.numel()
                stride = _round_up(cols, ALIGN)
                alloc_rows = 1
            self.layout.append((off, rows, cols, stride))
            off += alloc_rows * stride
        self.numel = _round_up(off, ALIGN)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.bf16 = torch.zeros(self.numel, dtype=torch.bfloat16, device=device)
        self.exp_avg = self.exp_avg_sq = None
        self.segment_ends = []
        for i, p in enumerate(order):
            view = self._view(self.flat, i, p)
            with torch.no_grad():
                view.copy_(p.data.to(device=device, dtype=torch.float32))
            p.data = view
            p.grad = self._view(self.grad, i, p)
            p._sam_bf16 = self._view(self.bf16, i, p)
            p._sam_flat = self
            p._sam_index = i
            if i + 1 == len(order) or self.group_of[i + 1] != self.group_of[i]:
                self.segment_ends.append(self.layout[i + 1][0] if i + 1 < len(order) else self.numel)
        self.refresh_shadows()

    def _view(self, buf, i, p):
        off, rows, cols, stride = self.layout[i]
        v = buf[off: off + rows * stride].view(rows, stride)[:, :cols]
        return v if p.dim() == 2 else v.reshape(p.shape) if stride == cols else v[0].view(p.shape)

    # ---- bf16 shadows --------------------------------------------------------------------------
    def refresh_shadows(self):
        ops.cast_bf16(self.flat, self.bf16)
        self._versions = [p._version for p in self.params]
        self.shadow_epoch = getattr(self, "shadow_epoch", 0) + 1      # consumers that keep derived copies of the shadows (decoder: re-tiled weights) compare it

    def ensure_fresh(self):
        """parameters modified in place from Python (load_state_dict, init) bump their version counter; the fused
        Adam kernel refreshes the shadows itself and does not."""
        if any(p._version != v for p, v in zip(self.params, self._versions)):
            self.refresh_shadows()

    def zero_grad(self, skip=None):
        """zero the gradient buffer; `skip`: sorted, disjoint [lo, hi) ranges to leave alone (gradients their producers overwrite, Trainer)"""
        if not skip:
            self.grad.zero_()
            return
        pos, pieces = 0, []
        for lo, hi in skip:
            if lo > pos:
                pieces.append((pos, lo))
            pos = max(pos, hi)
        if pos < self.numel:
            pieces.append((pos, self.numel))
        if self.grad.is_cuda and pieces and all(a % 4 == 0 and (b - a) % 4 == 0 for a, b in pieces):
            from . import ops
            ops.copy_blocks([(None, self.grad[a:b].view(1, 1, b - a)) for a, b in pieces])      # every range in one launch
        else:
            for a, b in pieces:
                self.grad[a:b].zero_()

    def range_of(self, module):
        """[start, end) element range of the flat buffers covering `module`'s parameters (they are contiguous)"""
        idx = sorted(p._sam_index for p in module.parameters())
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            raise ValueError("module parameters are not contiguous in flat storage")
        o0 = self.layout[idx[0]][0]
        last = idx[-1]
        return o0, (self.layout[last + 1][0] if last + 1 < len(self.layout) else self.numel)

    # ---- fused views -----------------------------------------------------------------------------
    @staticmethod
    def adjacent(ps, attr):
        """[3D, ...] view over three back-to-back parameters (query/key/value), or None if not adjacent"""
        ts = [getattr(p, attr) if attr else p.data for p in ps]
        t0 = ts[0]
        step = t0.numel() * t0.element_size() if t0.dim() == 1 else t0.shape[0] * t0.stride(0) * t0.element_size()
        for j, t in enumerate(ts):
            if t.data_ptr() != t0.data_ptr() + j * step or t.shape != t0.shape or t.stride() != t0.stride():
                return None
        if t0.dim() == 1:
            return torch.as_strided(t0, (len(ts) * t0.shape[0],), (1,))
        return torch.as_strided(t0, (len(ts) * t0.shape[0], t0.shape[1]), (t0.stride(0), 1))


def prepare(module, device=None, groups=None):
    """put `module`'s parameters into flat storage on the GPU (idempotent); returns the FlatParams"""
    fp = getattr(module, "_sam_flat_params", None)
    if fp is not None:
        return fp
    for b in module.buffers():
        b.data = b.data.to(device or "cuda")
    fp = FlatParams(module, device, groups)
    module._sam_flat_params = fp
    return fp


def flat_of(p):
    fp = getattr(p, "_sam_flat", None)
    if fp is None:
        raise RuntimeError("parameter is not in flat HIP storage: call sam_textvqa_amd.prepare(model) first")
    return fp
