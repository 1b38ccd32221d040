"""where the persistent decoding kernel and the per-kernel step disagree:  python tools/debug/decode_diff.py T OBJ OCR DEC"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from sam_textvqa_amd.params import prepare  # noqa: E402
from tests.test_decode_gpu import _batch  # noqa: E402
from tests.test_model_gpu import _small_full_model  # noqa: E402

shapes = tuple(int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (20, 160, 70, 20)
model, ref = _small_full_model(3, ("n", "s", "s"), shapes, vocab=300)
model.cuda().eval()
prepare(model)
model.decode_cache = True
outs = {}
for fused in ("0", "1"):
    os.environ["SAM_DECODE_FUSED"] = fused
    os.environ["SAM_DECODE_GRAPH"] = "1"
    model.__dict__.pop("_sam_decode_sessions", None)
    bd = _batch(5, shapes, 300, 41, "cuda")
    with torch.no_grad():
        sc = model(bd)["textvqa_scores"]
    outs[fused] = sc.float().cpu()
a, b = outs["0"], outs["1"]
live = a > -9000
d = ((a - b).abs() * live)
print("shape", tuple(a.shape), "max |a|", a[live].abs().max().item(), "max diff", d.max().item())
print("per step max diff:", [round(x, 4) for x in d.amax(dim=(0, 2)).tolist()])
print("per sample max diff:", [round(x, 4) for x in d.amax(dim=(1, 2)).tolist()])
col = d.amax(dim=(0, 1))
print("fixed-vocab columns max diff %.4f, OCR columns: %s" % (col[:300].max().item(), [round(x, 3) for x in col[300:].tolist()]))
from sam_textvqa_amd.synthetic import clone_batch  # noqa: E402
with torch.no_grad():
    want = ref.eval()(clone_batch(_batch(5, shapes, 300, 41, "cpu")))["textvqa_scores"].float()
for k, v in (("per-kernel", a), ("persistent", b)):
    e = ((v - want).abs() * live)
    print("%s vs fp32 oracle: max diff %.4f, per step %s" % (k, e.max().item(), [round(x, 3) for x in e.amax(dim=(0, 2)).tolist()]))
