"""CPU, gloo, world_size 2: the bucketed gradient reducer averages per-rank gradients and walks buckets from the end
of the flat buffer in backward order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from sam_textvqa_amd.parallel import GradReducer, init_distributed
    r, _, w = init_distributed()
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    n = 1000
    grad = (torch.arange(n, dtype=torch.float32) + 1) * (rank + 1) / world      # pre-scaled by 1/world, like the loss kernel does
    red = GradReducer(grad, bucket_bytes=4 * 300)                                # 300-element buckets -> 4 buckets
    assert [b for b in red.buckets] == [(700, 1000), (400, 700), (100, 400), (0, 100)]
    red.begin_step()
    red.region_done(750)           # nothing complete yet
    assert red.next_bucket == 0
    red.region_done(400)           # buckets 0 and 1 are final
    assert red.next_bucket == 2
    red.finish()
    expect = (torch.arange(n, dtype=torch.float32) + 1) * sum(range(1, world + 1)) / world
    q.put((rank, torch.allclose(grad, expect), red.next_bucket))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True, 4), (1, True, 4)]


def _bf16_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from sam_textvqa_amd.parallel import GradReducer, init_distributed
    init_distributed()
    n = 1001                                                                      # not a multiple of the world size: padded slices
    g = torch.Generator().manual_seed(7 + rank)
    mine = torch.randn(n, generator=g)
    grad = mine.clone()
    red = GradReducer(grad, bucket_bytes=4 * 300, payload="bf16")
    red.begin_step(); red.region_done(0); red.finish()
    all_ = [torch.randn(n, generator=torch.Generator().manual_seed(7 + r)) for r in range(world)]
    # what the wire format promises: every rank's contribution rounded to bf16, summed in fp32 in rank order, the sum rounded to bf16 once
    expect = sum(a.to(torch.bfloat16).float() for a in all_).to(torch.bfloat16).float()
    exact = sum(all_)
    q.put((rank, torch.equal(grad, expect), float((grad - exact).abs().max() / exact.abs().max()), grad.double().sum().item()))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_bf16_payload_gloo_world2():
    """SAM_GRAD_PAYLOAD=bf16: all-to-all of bf16 slices + fp32 sum on receipt + all-gather; both ranks end with the same bits"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bf16_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], res
    assert res[0][2] < 2.0 ** -7 and res[0][3] == res[1][3], res                   # bf16-accurate, replicas bit-identical


def test_single_process_reducer_is_noop():
    from sam_textvqa_amd.parallel import GradReducer
    g = torch.ones(10)
    red = GradReducer(g, bucket_bytes=16)
    red.region_done(0); red.finish()
    assert red.world_size == 1 and torch.equal(g, torch.ones(10))


def _sparse_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from sam_textvqa_amd.parallel import GradReducer, init_distributed
    init_distributed()
    rows_tab, d, dense = 50, 8, 300
    grad = torch.zeros(rows_tab * d + dense)
    grad[rows_tab * d:] = (rank + 1.0) / world                       # dense part: pre-scaled per-rank gradients
    def cpu_scatter(table, ids, rows, padding_idx):          # the package only has the HIP scatter; the exchange logic is what is tested here
        keep = (ids != padding_idx) & (ids >= 0) & (ids < table.shape[0])     # (sam_embedding_bwd_sorted skips out-of-range rows the same way)
        table.index_add_(0, ids[keep], rows[keep].to(table.dtype))
    red = GradReducer(grad, bucket_bytes=4 * 128, dense_lo=rows_tab * d, scatter_fn=cpu_scatter)
    assert red.buckets[-1][0] == rows_tab * d and all(lo >= rows_tab * d for lo, _ in red.buckets)   # the table is in no dense bucket
    g = torch.Generator().manual_seed(100 + rank)
    ids = torch.randint(0, rows_tab, (12,), generator=g)
    ids[:3] = 0                                                        # padding rows: skipped
    ids[3:5] = 7                                                       # a row both ranks (and one rank twice) touch
    rows = torch.randn(12, d, generator=g).to(torch.bfloat16)
    table = grad[: rows_tab * d].view(rows_tab, d)
    red.begin_step()
    red.sparse_rows(table, ids, rows, padding_idx=0)
    red.finish()
    # expectation: every rank's rows scattered, in any order
    exp = torch.zeros(rows_tab, d)
    for r in range(world):
        gr = torch.Generator().manual_seed(100 + r)
        i = torch.randint(0, rows_tab, (12,), generator=gr); i[:3] = 0; i[3:5] = 7
        v = torch.randn(12, d, generator=gr).to(torch.bfloat16).float()
        exp.index_add_(0, i[3:], v[3:])
    ok = torch.allclose(table, exp, atol=1e-6) and bool((table[0] == 0).all()) and torch.allclose(grad[rows_tab * d:], torch.full((dense,), sum(range(1, world + 1)) / world))
    # second step, uneven last batch: rank 1 holds fewer rows than were verified on the first call; it pads (no extra collective that rank 0 would
    # not enter) and both ranks still end with everyone's rows
    table.zero_()
    n2 = 12 if rank == 0 else 5
    ids2 = torch.arange(1, n2 + 1) + 10 * rank
    rows2 = torch.full((n2, d), float(rank + 1)).to(torch.bfloat16)
    red.begin_step()
    red.sparse_rows(table, ids2, rows2, padding_idx=0)
    red.finish()
    exp2 = torch.zeros(rows_tab, d)
    exp2.index_add_(0, torch.arange(1, 13), torch.full((12, d), 1.0))
    exp2.index_add_(0, torch.arange(1, 6) + 10, torch.full((5, d), 2.0))
    ok = ok and torch.equal(table, exp2)
    grew = False
    if rank == 0:                                              # a batch that GROWS is refused locally, before any collective
        try:
            red.sparse_rows(table, torch.arange(13), torch.zeros(13, d, dtype=torch.bfloat16), padding_idx=0)
        except RuntimeError:
            grew = True
    q.put((rank, ok and (grew or rank != 0)))
    dist.barrier()
    dist.destroy_process_group()


def test_sparse_table_exchange_gloo_world2():
    """row-sparse exchange of the word-embedding gradient: all ranks end with the sum of everyone's rows, the dense buckets skip the table"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sparse_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_global_loss_normaliser_identity():
    """Trainer.step's data-parallel scaling: sum_r (c_r / C) * mean_r(grad) == global mean gradient, for unequal per-rank counts"""
    g = torch.Generator().manual_seed(0)
    per_rank = [torch.randn(n, 5, generator=g) for n in (7, 1, 12)]            # per-sample "gradients" of the unmasked steps of 3 ranks
    counts = torch.tensor([float(len(x)) for x in per_rank])
    total = counts.sum()
    ddp = sum((c / total) * x.mean(0) for c, x in zip(counts, per_rank))
    assert torch.allclose(ddp, torch.cat(per_rank).mean(0), atol=1e-6)


def test_rccl_binding_resolves_every_entry_point_and_declines_without_a_group():
    """sam_textvqa_amd/rccl.py: the library torch.distributed's "nccl" backend is built on is found among the process's mapped objects (or next to torch), every
    entry point the reducer calls resolves with the rccl.h signatures, and without an RCCL process group there is no communicator: callers stay on the
    process-group API (gloo: these CPU tests and the shared-GPU two-rank tests)."""
    from sam_textvqa_amd import rccl
    if rccl.library_path() is None:
        pytest.skip("no librccl.so mapped into this process or next to torch (CPU-only / non-ROCm torch)")
    l = rccl.lib()
    for name in ("ncclAllReduce", "ncclAllGather", "ncclAllToAll", "ncclBroadcast", "ncclCommCount", "ncclGetErrorString"):
        assert hasattr(l, name), name
    assert l.ncclGetErrorString(0) == b"no error"
    assert rccl.communicator() is None
    assert rccl._DTYPES[torch.float32] == 7 and rccl._DTYPES[torch.bfloat16] == 9 and rccl._DTYPES[torch.int64] == 4 and (rccl.SUM, rccl.MAX) == (0, 2)      # rccl.h enums


def _agree_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from sam_textvqa_amd import parallel
    parallel.init_distributed()
    red = parallel.GradReducer(torch.zeros(64), bucket_bytes=4 * 16)
    assert red.comm is None                                   # gloo: the process-group transport
    t = torch.tensor([float(rank + 1)])
    wait = red.reduce_scalar(t)                               # the global loss-normaliser count of Trainer._eager_step
    wait()
    buf = torch.full((8,), float(rank))
    red.broadcast(buf, src=1)
    out = (parallel.agree(True), parallel.agree(rank == 0), parallel.agree(False), float(t.item()), buf.tolist())
    parallel.quiesce_before_capture()                         # no-op off the RCCL backend
    q.put((rank,) + out)
    dist.barrier()
    dist.destroy_process_group()


def test_capture_agreement_and_scalar_reduce_gloo_world2():
    """a hipGraph capture that fails on ONE rank must send ALL ranks down the eager path (parallel.agree: MIN over the ranks, host-visible); the reducer's
    scalar all-reduce and start-up broadcast on the process-group transport"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True, False, False, 3.0, [1.0] * 8), (1, True, False, False, 3.0, [1.0] * 8)]
    from sam_textvqa_amd import parallel
    assert parallel.agree(True) is True and parallel.agree(False) is False      # no group: the local answer
    assert parallel.CAPTURE_ERROR_MODE == "thread_local"


def test_buckets_are_cut_at_every_region_boundary(monkeypatch):
    """round 6: one bucket per finality region (a region larger than bucket_bytes is subdivided), so a layer's gradients leave when THAT layer is final -- a bucket
    spanning 2.3 layers used to wait for all of them.  Host logic only: no process group, no GPU."""
    from sam_textvqa_amd.parallel import GradReducer
    grad = torch.zeros(1000)
    red = GradReducer(grad, bucket_bytes=4 * 300, sparse_range=(100, 250))
    ids = red.register_regions([(820, 1000), (640, 820), (250, 640), (40, 100)])          # the row-sparse table sits between the last two
    assert ids == [0, 1, 2, 3]
    assert red.buckets == [(820, 1000), (640, 820), (340, 640), (250, 340), (40, 100), (0, 40)]
    red.begin_step()
    red.mark_done(1)                       # out of order: nothing leaves before region 0 is final
    assert red.next_bucket == 0
    red.mark_done(0)
    assert red.next_bucket == 2            # ... then both regions' buckets
    red.mark_done(2)
    assert red.next_bucket == 4            # the subdivided region: both of its buckets
    red.mark_done(3)
    assert red.next_bucket == 5
    red.finish()
    assert red.late_buckets == 1           # [0, 40) has no region: it leaves at finish()
    monkeypatch.setenv("SAM_BUCKET_PER_REGION", "0")
    old = GradReducer(torch.zeros(1000), bucket_bytes=4 * 300, sparse_range=(100, 250))
    old.register_regions([(820, 1000), (640, 820), (250, 640), (40, 100)])
    assert old.buckets == [(700, 1000), (400, 700), (250, 400), (40, 100), (0, 40)]      # rounds 2-5: only the low end of the regions is a forced cut
