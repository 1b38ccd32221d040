"""Data-parallel gradient exchange: one process per GPU, torch.distributed backend "nccl" (= RCCL on ROCm) over xGMI.

The reference's only strategy is single-process nn.DataParallel (train.py:111-112: per-step parameter broadcast +
gradient reduce to GPU0).  Here every rank keeps a full replica (same seed-0 weights), shards the batch, and the flat
fp32 gradient buffer (params.py) is all-reduced in a few large contiguous buckets — no per-parameter messages, no
broadcast per step.  Buckets are walked from the END of the buffer (the last layers finish their backward first);
with overlap=True each bucket's all-reduce is issued on a side stream as soon as the layers covering it have run their
backward (EncoderLayerFn.backward calls `layer_done`), hiding the exchange behind the remaining backward GEMMs.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): large buckets (default 64 MB) keep RCCL's ring/tree per-link
bound instead of latency bound; 386 MB of fp32 gradients = 6 buckets.
The loss normaliser max(sum(loss_mask),1) is the GLOBAL batch's, as under the reference's DataParallel (task_utils.py:28-29): every rank
all-reduces its count (one float, underneath the forward pass) and scales its loss gradient by count_rank / count_global, so the all-reduce
SUM of the per-rank gradients is the gradient of the global mean (Trainer.step).

Sparse table: the 30522 x 768 word-embedding table is 94 MB of the 386 MB gradient buffer, touches at most B*20 rows per step and
is the LAST gradient of the backward pass (nothing left to hide its exchange behind).  When it sits at the bottom of the flat buffer
the reducer leaves it out of the dense buckets (`dense_lo`); instead every rank all-gathers the (row index, bf16 gradient row) pairs
(2 MB per rank) and scatters all ranks' rows into its own table gradient: same sum, 1/4 less all-reduce volume, and the exposed tail of the
exchange shrinks from 258 MB to 164 MB."""
import os
import time

import torch
import torch.distributed as dist

from . import rccl

CAPTURE_ERROR_MODE = "thread_local"
# CUs the backward's persistent grids leave to RCCL's channel kernels when the job spans more than one rank (Trainer; csrc/gemm_common.h: grid_cu_count), unless
# SAM_DP_CU_RESERVE says otherwise.  0 = off, the measured choice: on one GPU with the collectives emulated inside the captured step (tools/bench_cu_reserve.py,
# profiles/r6_cu_reserve.txt) a reserve of 16 / 32 costs the backward 0.2 ms per step (fewer CUs per launch, coarser tile rounds) and saves the same 0.2-0.3 ms
# (kernels beside a 32-channel collective run 1.3-1.6x longer for the ~0.9 ms per step that collectives and backward kernels overlap): no net gain.
DEFAULT_CU_RESERVE = 0


def quiesce_before_capture():
    """call before a stream capture is opened in a process that holds an RCCL process group.
    ProcessGroupNCCL's watchdog thread polls the end events of its outstanding Work objects (hipEventQuery, every 100 ms).  On this HIP runtime such a
    query from another thread (a) always fails under a GLOBAL-mode capture and (b) in every mode fails with hipErrorCapturedEvent when the stream the event
    was last recorded on is capturing now; either one invalidates the capture and the watchdog answers with std::terminate (rccl.py; round-4 driver record).
    The package therefore (1) captures in thread-local mode (CAPTURE_ERROR_MODE), (2) keeps its own collectives off the process-group API (rccl.py: no Work
    objects on the step's path) and (3) here lets the watchdog retire what rendezvous-time / bench-bracket collectives left behind: everything on the device
    completes, then the watchdog gets a few of its polling periods."""
    if not (dist.is_initialized() and torch.cuda.is_available()):
        return
    try:
        if dist.get_backend() != "nccl":
            return
    except Exception:
        return
    torch.cuda.synchronize()
    time.sleep(float(os.environ.get("SAM_CAPTURE_QUIESCE_S", "0.35")))


def agree(ok, group=None):
    """True iff `ok` holds on EVERY rank (a capture that failed on one rank must send all ranks down the eager path: half the ranks replaying a graph
    while the others enqueue the step's collectives one by one would deadlock).  Host-visible, through the process group, outside any capture."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(ok)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


class GradReducer:
    def __init__(self, flat_grad, bucket_bytes=64 << 20, overlap=True, group=None, dense_lo=0, scatter_fn=None, sparse_range=None, payload=None):
        self.grad = flat_grad
        # wire format of the dense buckets.  "fp32": one all-reduce per bucket.  "bf16" (SAM_GRAD_PAYLOAD=bf16): half the bytes on the links --
        # all-to-all of bf16 slices (rank j receives slice j of every rank: on xGMI's fully connected point-to-point links every pair carries
        # 1/W of the bucket at the same time), fp32 sum of the W slices on receipt, all-gather of the bf16-rounded sums.  Every rank ends
        # with the same bits (each slice is summed by exactly one rank), the sum itself is rounded to bf16 once.
        self.payload = payload or os.environ.get("SAM_GRAD_PAYLOAD", "fp32")
        if self.payload not in ("fp32", "bf16"):
            raise ValueError("GradReducer payload must be 'fp32' or 'bf16', not %r" % (self.payload,))
        self.scatter_fn = scatter_fn or _scatter_rows
        # [sparse_lo, sparse_hi) is exchanged through sparse_rows(), not all-reduced.  `dense_lo=k` is the short form of sparse_range=(0, k)
        # (two optimizer groups: the word-embedding table is the first parameter of flat storage); with the reference's three groups
        # (text_bert_init_from_bert_base) the table sits in the middle of the buffer, in front of the rest of TextBert.
        self.sparse_lo, self.sparse_hi = (0, int(dense_lo)) if sparse_range is None else (int(sparse_range[0]), int(sparse_range[1]))
        self.dense_lo = self.sparse_hi if self.sparse_lo == 0 else 0
        self._checked_rows = -1
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        # SAM_FORCE_DIST=1: run the collectives even in a 1-rank group (exercises the RCCL / side-stream path on a single GPU)
        self.force = dist.is_initialized() and os.environ.get("SAM_FORCE_DIST") == "1"
        self.per_bucket = max(1, bucket_bytes // flat_grad.element_size())
        self._build_buckets(cut=None)
        # SAM_REDUCER_CHECK=1 (1-rank groups only, where the all-reduce is the identity): keep a copy of every bucket as it is released and
        # verify at finish() that nothing wrote into it afterwards -- catches a premature release on a single GPU
        self.check = os.environ.get("SAM_REDUCER_CHECK") == "1" and self.world_size == 1
        self.overlap = overlap and flat_grad.is_cuda and (self.world_size > 1 or self.force)
        self.stream = torch.cuda.Stream() if self.overlap else None
        self.regions = []
        self.barrier_names, self.barrier_regions = set(), []
        self.comm = self._resolve_comm()
        # SAM_EMULATE_COMM="<channels>:<GB/s>" (1-rank groups only, a measurement aid): every collective of the step is followed, on the SAME stream, by a
        # stand-in for what it would be on N GPUs -- <channels> workgroups holding a CU each (ops.debug_cu_hog) for bytes / <GB/s> -- so that one GPU shows
        # what RCCL's channel kernels cost the persistent kernels they run beside, in the captured step's own stream structure (tools/bench_cu_reserve.py)
        self.emulate = None
        if os.environ.get("SAM_EMULATE_COMM") and self.world_size == 1 and self.grad.is_cuda:
            ch, gbps = os.environ["SAM_EMULATE_COMM"].split(":")
            self.emulate = (int(ch), float(gbps))
        self.begin_step()

    def _resolve_comm(self):
        """the group's ncclComm_t when the group is an RCCL group: the collectives are then enqueued directly on the reducer's stream (rccl.py) -- no
        ProcessGroupNCCL Work objects, no watchdog polling of the step's events.  None: gloo (tests), SAM_RCCL_DIRECT=0, a 1-rank job without a group.
        The choice is AGREED over the group: a rank whose lookup failed (rccl.communicator swallows local errors) would otherwise call
        dist.all_to_all_single while its peers call ncclAllToAll on the same communicator -- a hang.  Every rank runs the same collectives here, in the
        same order, whatever its own lookup returned: one warm-up all-reduce (the communicator is built lazily by the group's first collective; every
        rank constructs its reducer at the same point) and one MIN all-reduce of the "have it" flag."""
        if not (self.grad.is_cuda and dist.is_initialized() and (self.world_size > 1 or self.force)):
            return None
        try:
            is_rccl = dist.get_backend(self.group) == "nccl"
        except Exception:
            is_rccl = False
        if not is_rccl:
            return None                                    # (the backend is a property of the group: the same answer on every rank)
        t = torch.zeros(1, dtype=torch.float32, device=self.grad.device)
        dist.all_reduce(t, group=self.group)               # unconditional: builds the communicator where the group has not used it yet
        torch.cuda.synchronize()
        comm = rccl.communicator(self.group)               # (None under SAM_RCCL_DIRECT=0)
        have = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=self.grad.device)
        dist.all_reduce(have, op=dist.ReduceOp.MIN, group=self.group)
        if not int(have.item()):
            if comm is not None:
                import logging
                logging.getLogger(__name__).warning("another rank has no direct RCCL communicator: the whole group stays on torch.distributed calls")
            return None
        return comm

    def _build_buckets(self, cut):
        """buckets walk the dense ranges [sparse_hi, n) and [0, sparse_lo) from the end (ascending index = descending addresses = backward
        order).  `cut`: a forced boundary (the low end of the registered regions): a bucket straddling it would mix gradients that are final
        early (encoder layers) with ones that are final last (everything below) and could only leave at finish()."""
        n, per = self.grad.numel(), self.per_bucket
        cuts = set() if cut is None else ({cut} if isinstance(cut, int) else set(cut))
        edges = sorted({0, n, self.sparse_lo, self.sparse_hi} | {c for c in cuts if 0 < c < n}, reverse=True)
        self.buckets = []
        for top, bottom in zip(edges[:-1], edges[1:]):
            if bottom >= self.sparse_lo and top <= self.sparse_hi:
                continue                                   # the row-sparse table
            hi = top
            while hi > bottom:
                lo = max(bottom, hi - per)
                self.buckets.append((lo, hi))
                hi = lo

    def register_regions(self, ranges):
        """[lo, hi) flat ranges of the encoder layers; returns region ids in the order given.  A bucket is released once
        every registered region at or above its low end has reported `mark_done` (layers may finish in any order)."""
        order = sorted(range(len(ranges)), key=lambda i: -ranges[i][0])
        self.regions = [ranges[i] for i in order]
        for (lo_hi, above) in zip(self.regions[1:], self.regions[:-1]):
            if lo_hi[1] != above[0] and not (lo_hi[1] == self.sparse_lo and above[0] == self.sparse_hi):      # (the sparse table may sit between two regions)
                raise ValueError("gradient regions must tile a contiguous range up to the end of the buffer: gap between %r and %r" % (lo_hi, above))
        if self.regions and self.regions[0][1] != self.grad.numel():
            raise ValueError("the highest gradient region must end at the end of the flat buffer")
        if self.regions:
            # Every region boundary is a bucket boundary (SAM_BUCKET_PER_REGION=0: only the low end of the regions, as rounds 2-5 had it).  A 64 MB bucket
            # spans 2.3 encoder layers (28.3 MB each) and leaves only when ALL of them are final -- with the weight gradients going out in layer pairs the first
            # bucket left after FOUR of the MMT's six layers (profiles/r6_dp_timeline.txt: 1.9 ms into a 3.8 ms backward) and three of five buckets after
            # its end.  One bucket per region leaves with its region: 12 collectives of <= 28 MB instead of 5 of 64 MB, the first two at the first pair.
            per_region = os.environ.get("SAM_BUCKET_PER_REGION", "1") != "0"
            self._build_buckets(cut={lo for lo, _ in self.regions} if per_region else min(lo for lo, _ in self.regions))
        ids = [0] * len(ranges)
        for pos, i in enumerate(order):
            ids[i] = pos
        self.begin_step()
        return ids

    def _note_stream(self):
        if self.grad.is_cuda:
            cur = torch.cuda.current_stream()
            if all(cur != st for st in self.grad_streams):
                self.grad_streams.append(cur)

    def mark_done(self, region_id):
        self._note_stream()
        self.done[region_id] = True
        while self.done_ptr < len(self.regions) and self.done[self.done_ptr]:
            self.done_ptr += 1
        if self.done_ptr > 0:
            self.region_done(self.regions[self.done_ptr - 1][0])

    def set_barrier(self, names, region_ids):
        """the regions `region_ids` (one id or a list) are final once every name in `names` has been reported by barrier_hit
        (autograd.GradBarrierFn)"""
        self.barrier_names = set(names)
        self.barrier_regions = [region_ids] if isinstance(region_ids, int) else list(region_ids)

    @property
    def barrier_region(self):
        return self.barrier_regions[0] if self.barrier_regions else None

    def barrier_hit(self, name):
        if not self.barrier_regions:
            return
        self._note_stream()
        self.barrier_seen.add(name)
        if self.barrier_names <= self.barrier_seen:
            for rid in self.barrier_regions:
                if not self.done[rid]:
                    self.mark_done(rid)

    def begin_step(self):
        self.grad_streams = [torch.cuda.current_stream()] if self.grad.is_cuda else []
        self.next_bucket = 0
        self.ready_lo = self.grad.numel()      # gradients at addresses >= ready_lo are final
        self.work = []
        self._keep = []
        self.snapshots = []
        self.barrier_seen = set()
        self.done = [False] * len(self.regions)
        self.done_ptr = 0

    def _launch(self, k):
        lo, hi = self.buckets[k]
        chunk = self.grad[lo:hi]
        if self.world_size == 1 and not self.force:
            return
        if self.overlap:
            # gradients are written on the stream the step started on and (TextBert) on a side stream, and a release may be triggered from
            # either one for regions finished on the other: wait for every stream a finality mark was reported from -- everything the
            # bucket needs was ENQUEUED before this call (host order), this makes it COMPLETE
            self._note_stream()
            for st in self.grad_streams:
                self.stream.wait_stream(st)
            with torch.cuda.stream(self.stream):
                if self.check:
                    self.snapshots.append((k, self._snapshot(chunk)))
                self._exchange(chunk, True)
        else:
            if self.check:
                self.snapshots.append((k, self._snapshot(chunk)))
            self._exchange(chunk, False)

    def _snapshot(self, chunk):
        """what a 1-rank exchange must leave in the bucket (SAM_REDUCER_CHECK): the bucket itself, bf16-rounded under the bf16 payload"""
        return chunk.clone() if self.payload == "fp32" else chunk.to(torch.bfloat16).float()

    def _exchange(self, chunk, async_op):
        """sum `chunk` over the ranks, in place"""
        if self.payload == "fp32":
            if self.comm is not None:
                rccl.all_reduce(self.comm, chunk)                      # on the current stream: the reducer's own under overlap, the step's otherwise
                self._emulated(chunk.numel() * chunk.element_size())
                return
            w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if async_op:
                self.work.append(w)
            return
        W, n = self.world_size, chunk.numel()
        per = (n + W - 1) // W
        send = torch.zeros((W * per,), dtype=torch.bfloat16, device=chunk.device)
        send[:n] = chunk                                               # fp32 -> bf16 (RNE), zero tail
        recv = torch.empty_like(send)
        if self.comm is not None:
            rccl.all_to_all(self.comm, recv, send, W)
        else:
            dist.all_to_all_single(recv, send, group=self.group)       # recv[r * per : (r + 1) * per] = rank r's copy of MY slice
        mine = recv.view(W, per).float().sum(dim=0).to(torch.bfloat16)  # fp32 accumulation over the ranks, in rank order
        out = torch.empty_like(send)
        self._all_gather(out, mine)
        chunk.copy_(out[:n])
        self._keep.append((send, recv, mine, out))                     # (allocated on the caller's stream, used on the reducer's: held until finish())

    def _emulated(self, nbytes):
        if self.emulate is not None:
            from . import ops
            ops.debug_cu_hog(self.emulate[0], nbytes / (self.emulate[1] * 1e3))      # (GB/s = bytes per ns * 1; us = bytes / (GB/s * 1e3))

    def _all_gather(self, out, t):
        if self.comm is not None:
            rccl.all_gather(self.comm, out, t)
        else:
            dist.all_gather_into_tensor(out, t, group=self.group)

    def reduce_scalar(self, t):
        """t (a small device tensor) := its sum over the ranks, started now and running underneath whatever the caller enqueues next; returns wait(): call
        it on the stream that is about to read t.  (The loss normaliser: the count of unmasked decoding steps of the GLOBAL batch, Trainer._eager_step.)"""
        if self.comm is None:
            return dist.all_reduce(t, group=self.group, async_op=True).wait
        if not self.overlap:
            rccl.all_reduce(self.comm, t)
            return lambda: None
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            rccl.all_reduce(self.comm, t)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return lambda: torch.cuda.current_stream().wait_event(ev)

    def broadcast(self, t, src=0):
        """rank `src`'s t to every rank, complete on return (start-up: masters and optimizer state)"""
        if self.comm is not None:
            # `src` is a GLOBAL rank (torch.distributed's convention); ncclBroadcast wants the rank inside the communicator, i.e. inside the group
            root = dist.get_group_rank(self.group, src) if self.group is not None else src
            rccl.broadcast(self.comm, t.view(-1) if t.is_contiguous() else t, root)
            torch.cuda.synchronize()
        else:
            dist.broadcast(t, src=src, group=self.group)

    def region_done(self, lo):
        """everything at flat offsets >= lo has its final gradient: launch every bucket that is now complete"""
        self.ready_lo = min(self.ready_lo, lo)
        while self.next_bucket < len(self.buckets) and self.buckets[self.next_bucket][0] >= self.ready_lo:
            self._launch(self.next_bucket)
            self.next_bucket += 1

    def sparse_rows(self, grad_table, ids, rows, padding_idx=-1):
        """grad_table[ids[t], :] += rows[t, :] for the rows of EVERY rank (ids int64 [R], rows bf16/fp32 [R, D], same R on all ranks):
        the data-parallel exchange of a row-sparse gradient living in [0, dense_lo)."""
        if self.world_size > 1 or self.force:
            w = self.world_size
            if self._checked_rows < 0:                  # all_gather_into_tensor needs the same row count on every rank: verified, loudly, on
                                                        # the first call (every rank makes it: no rank can skip the collective)
                cnt = torch.tensor([ids.numel(), -ids.numel()], dtype=torch.int64, device=ids.device)
                if self.comm is not None:
                    rccl.all_reduce(self.comm, cnt, op=rccl.MAX)
                else:
                    dist.all_reduce(cnt, op=dist.ReduceOp.MAX, group=self.group)
                if int(cnt[0]) != -int(cnt[1]):
                    raise RuntimeError("GradReducer.sparse_rows: ranks hold different numbers of rows (%d..%d); pad the last batch" % (-int(cnt[1]), int(cnt[0])))
                self._checked_rows = ids.numel()
            elif ids.numel() < self._checked_rows:
                # a SHORTER list later on (uneven last batch on this rank): padded up to the verified count with out-of-range indices, which the
                # scatter skips -- every rank still gathers equal-sized pieces and no extra collective (that only some ranks would enter) is needed
                pad = self._checked_rows - ids.numel()
                ids = torch.cat([ids.reshape(-1), torch.full((pad,), -1, dtype=ids.dtype, device=ids.device)])
                rows = torch.cat([rows, rows.new_zeros((pad, rows.shape[1]))])
            elif ids.numel() > self._checked_rows:
                raise RuntimeError("GradReducer.sparse_rows: %d rows on this rank, %d were verified equal across ranks on the first step; "
                                   "batches may shrink (they are padded) but not grow" % (ids.numel(), self._checked_rows))
            ids, rows = ids.contiguous(), rows.contiguous()
            ids_all = torch.empty((w * ids.numel(),), dtype=ids.dtype, device=ids.device)
            rows_all = torch.empty((w * rows.shape[0], rows.shape[1]), dtype=rows.dtype, device=rows.device)
            if self.overlap:
                # gather + sort + scatter leave the compute stream: the table's rows are ready (enqueued) when this is called, whatever is
                # left of the backward pass -- the object / OCR encoders' weight gradients -- runs next to the exchange; finish() joins
                self._note_stream()
                self.stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.stream):
                    self._all_gather(ids_all, ids)
                    self._all_gather(rows_all, rows)
                    self.scatter_fn(grad_table, ids_all, rows_all, padding_idx)
                self._keep.append((ids, rows, ids_all, rows_all))
                return
            self._all_gather(ids_all, ids)
            self._all_gather(rows_all, rows)
            ids, rows = ids_all, rows_all
        self.scatter_fn(grad_table, ids, rows, padding_idx)

    def finish(self):
        """after backward: reduce whatever is left, then make the compute stream wait for the exchange"""
        self.late_buckets = len(self.buckets) - self.next_bucket        # buckets no finality mark released (0 when the regions tile the whole buffer)
        self.region_done(0)
        for w in self.work:
            w.wait()
        if self.overlap:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.work, self._keep = [], []
        for k, snap in self.snapshots:
            lo, hi = self.buckets[k]
            if not torch.equal(snap, self.grad[lo:hi]):
                bad = (snap != self.grad[lo:hi]).nonzero()
                raise RuntimeError("GradReducer: bucket %d [%d, %d) was released before its gradients were final (first late write at offset %d)"
                                   % (k, lo, hi, lo + int(bad[0])))
        self.snapshots = []


def _scatter_rows(grad_table, ids, rows, padding_idx):
    """grad_table[ids[t], :] += rows[t, :] on the GPU, in a FIXED order: every table row is summed by ONE thread block -- the block of its first occurrence in
    the list, which adds the later duplicates in list order (sam_embedding_bwd: no atomics, no sort) -- so all ranks, which hold the same gathered
    (index, row) list, add up bit-identical gradients and the replicas stay in lock-step (atomics would sum in an arbitrary order per rank).
    There is no CPU implementation in the package: the gloo unit test of GradReducer injects its own `scatter_fn`."""
    from . import ops
    # sam_embedding_bwd takes its fixed-order (first-occurrence owner) kernel only for 16-byte aligned tables whose row stride is a multiple of 4 floats,
    # and not under SAM_EMBED_BWD_ATOMIC=1; otherwise it falls back to fp32 atomics, whose order differs from rank to rank: replicas would drift with no
    # diagnostic.  Under the data-parallel exchange that is an error, not a fallback.
    if os.environ.get("SAM_EMBED_BWD_ATOMIC") == "1" or grad_table.stride(0) % 4 or grad_table.data_ptr() % 16:
        raise RuntimeError("data-parallel row-sparse scatter needs the deterministic sam_embedding_bwd kernel: table row stride %d (must be a multiple of 4), "
                           "address %% 16 = %d (must be 0), SAM_EMBED_BWD_ATOMIC=%s (must be unset)"
                           % (grad_table.stride(0), grad_table.data_ptr() % 16, os.environ.get("SAM_EMBED_BWD_ATOMIC")))
    ops.embedding_bwd((rows if rows.dtype == torch.bfloat16 else rows.to(torch.bfloat16)).contiguous(), ids.contiguous(), grad_table, padding_idx)


active_reducer = None   # set by the Trainer; EncoderLayerFn.backward reports finished layers to it


def init_distributed():
    """read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run contract)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("SAM_FORCE_DIST") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        # SAM_DIST_BACKEND=gloo + SAM_DIST_SHARE_GPU=1: several ranks on ONE GPU through gloo's CUDA-tensor collectives -- how the multi-rank
        # logic (bucket release, row-sparse exchange, global loss normaliser, replica consistency) is tested end to end on a 1-GPU box
        backend = os.environ.get("SAM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if os.environ.get("SAM_DIST_SHARE_GPU") == "1":
            local = 0
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        # (RCCL's channel count is left to RCCL: NCCL_MAX_NCHANNELS / NCCL_MIN_NCHANNELS are the user's; a reserve -- SAM_DP_CU_RESERVE -- should not be smaller
        # than the channels RCCL ends up using, or the blocks beyond it displace persistent blocks all the same.  DESIGN.md section 6.)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world
