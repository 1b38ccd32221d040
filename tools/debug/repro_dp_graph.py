"""Run the captured-RCCL-step child of tests/test_zz_ddp_gpu.py N times and keep the FULL stdout / stderr of every run that does not end clean
(round-4 driver record: the child died with SIGABRT on a non-Python thread; pytest's repr cut the cause).
usage: python tools/debug/repro_dp_graph.py [N] [out_dir]"""
import os
import socket
import subprocess
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "gpurun_out", "dp_repro")
os.makedirs(out, exist_ok=True)
import importlib
mod = None
for name in ("tests.test_zz_ddp_gpu", "tests.test_ddp_gpu"):
    try:
        mod = importlib.import_module(name)
        break
    except ImportError:
        pass
script = mod._GRAPH_DP_SCRIPT
bad = 0
for i in range(n):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SAM_REPO=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               TORCH_SHOW_CPP_STACKTRACES="1")
    env.pop("SAM_REDUCER_CHECK", None)
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        rc, so, se = r.returncode, r.stdout, r.stderr
    except subprocess.TimeoutExpired as e:
        rc, so, se = "timeout", (e.stdout or b"").decode("utf-8", "replace") if isinstance(e.stdout, bytes) else (e.stdout or ""), (e.stderr or b"").decode("utf-8", "replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
    ok = rc == 0 and "GRAPH_DP_OK" in so
    print("run %2d rc %s ok %s %.1fs" % (i, rc, ok, time.time() - t0), flush=True)
    if not ok:
        bad += 1
        with open(os.path.join(out, "fail_%02d.stdout" % i), "w") as f:
            f.write(so)
        with open(os.path.join(out, "fail_%02d.stderr" % i), "w") as f:
            f.write(se)
print("DP_REPRO %d/%d clean" % (n - bad, n))
sys.exit(1 if bad else 0)
