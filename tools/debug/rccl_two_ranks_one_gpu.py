#!/usr/bin/env python3
"""Probe: does RCCL accept two ranks on ONE device?  (It would let the driver's N > 1 command run through the captured, rccl-direct step on a 1-GPU box.)
    python tools/debug/rccl_two_ranks_one_gpu.py          -> prints RCCL_2ON1 ok / the error of each rank"""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp


def _run(rank, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", 0))
        t = torch.full((1024,), float(rank + 1), device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        q.put((rank, "ok", float(t[0].item())))
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, "error", "%s: %s" % (type(e).__name__, str(e)[:400])))


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_run, args=(r, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = []
    for p in ps:
        p.join(120)
        if p.is_alive():
            p.kill()
            res.append(("?", "hung", ""))
    while not q.empty():
        res.append(q.get())
    print("RCCL_2ON1", sorted(res, key=str))
