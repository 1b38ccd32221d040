"""CPU, gloo, world_size 2: the bucketed gradient reducer averages per-rank gradients and walks buckets from the end
of the flat buffer in backward order."""
import os
import socket

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


def test_single_process_reducer_is_noop():
    from sam_textvqa_amd.parallel import GradReducer
    g = torch.ones(10)
    red = GradReducer(g, bucket_bytes=16)
    red.region_done(0); red.finish()
    assert red.world_size == 1 and torch.equal(g, torch.ones(10))
