import os, sys, torch
os.environ.setdefault("SAM_DECODE_GRAPH", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_model
from sam_textvqa_amd.params import prepare
from sam_textvqa_amd.synthetic import make_batch
model = build_model(3, ("n", "n", "s", "s", "s", "s"), 5000).cuda().eval()
prepare(model)
model.decode_cache = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bd = make_batch(B, device="cuda", seed=1)
try:
    with torch.no_grad():
        model(bd)
except Exception as e:
    print("EXC", str(e)[:100])
torch.cuda.synchronize()
ses = next(iter(model._sam_decode_sessions.values()))
w = ses._fused_ws
print("err", int(w[256]), "bars", [(int(w[g * 32]), int(w[g * 32 + 1])) for g in range(8)])
print("prev", ses.prev[:2].tolist())
