import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.full((1000,), float(rank + 1), device="cuda:0")
    dist.all_reduce(t)
    ok1 = bool((t == 3).all())
    src = torch.arange(4, device="cuda:0", dtype=torch.int64) + 10 * rank
    out = torch.empty(8, device="cuda:0", dtype=torch.int64)
    try:
        dist.all_gather_into_tensor(out, src); ok2 = out.tolist()
    except Exception as e:
        ok2 = "all_gather_into_tensor: %r" % e
    b = torch.ones(16, 8, device="cuda:0", dtype=torch.bfloat16) * (rank + 1)
    ob = torch.empty(32, 8, device="cuda:0", dtype=torch.bfloat16)
    try:
        dist.all_gather_into_tensor(ob, b); ok3 = ob.float().sum().item()
    except Exception as e:
        ok3 = "bf16 gather: %r" % e
    wk = dist.all_reduce(torch.ones(3, device="cuda:0"), async_op=True); wk.wait()
    print("rank", rank, ok1, ok2, ok3, flush=True)
    dist.barrier(); dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(w, args=(2, 29611), nprocs=2)
