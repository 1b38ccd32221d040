"""per-phase time of the persistent greedy-decoding kernel (SAM_DECODE_PROF=1: block 0 stamps the 100 MHz wall clock after every grid barrier)"""
import os, sys, torch
os.environ["SAM_DECODE_PROF"] = "1"
os.environ.setdefault("SAM_DECODE_GRAPH", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_model
from sam_textvqa_amd.params import prepare
from sam_textvqa_amd.synthetic import make_batch
model = build_model(3, ("n", "n", "s", "s", "s", "s"), 5000).cuda().eval()
prepare(model)
model.decode_cache = True
for _ in range(3):
    bd = make_batch(64, device="cuda", seed=1)
    with torch.no_grad():
        model(bd)
torch.cuda.synchronize()
ses = next(iter(model._sam_decode_sessions.values()))
st = ses._fused_ws[288:288 + 2048].view(torch.int64).cpu().tolist()
n_layers = 6
names = ["E0"] + (["Q", "A", "O", "F1", "G", "H", "F2"] * n_layers + ["C", "P"]) * 11
ts = [st[k] for k in range(1, len(names) + 1)]
print("total us", (ts[-1] - ts[0]) / 100.0, "phases", len(names) - 1)
agg = {}
for k in range(1, len(names)):
    agg.setdefault(names[k], []).append((ts[k] - ts[k - 1]) / 100.0)
for n, v in agg.items():
    print("%-3s n=%3d mean %.2f us  min %.2f max %.2f" % (n, len(v), sum(v) / len(v), min(v), max(v)))
step = 7 * n_layers + 2
print("per step us:", [round((ts[step * (i + 1)] - ts[step * i]) / 100.0, 1) for i in range(11)])
