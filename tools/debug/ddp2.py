import os, sys, traceback, torch, torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
def w(rank, world, port):
    try:
        from tests.test_ddp_gpu import _run
        _run(rank, world, port, "/tmp")
        x = torch.load("/tmp/w%d_r%d.pt" % (world, rank))
        print("rank", x[0], "losses", x[1], "psum", float(x[2].double().sum()), flush=True)
    except Exception:
        traceback.print_exc(); sys.stdout.flush(); raise
if __name__ == "__main__":
    world = int(sys.argv[1])
    mp.spawn(w, args=(world, 29711 + world), nprocs=world)
