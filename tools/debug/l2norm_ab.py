"""debug: hashes of the l2norm outputs of the beam test's batch (plain and beam-expanded), with guard rows around the output: equal hashes = bit-identical kernels (used to compare builds of embed.hip)"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_decode_gpu import _models, _batch
from sam_textvqa_amd import ops
model, ref, shapes = _models(layers=("n", "s"))
bd = _batch(6, shapes, 300, 31, "cpu")
def pack(parts, nz):
    b, n = parts[0].shape[:2]
    k = sum(p.shape[-1] for p in parts) + nz
    kp = (k + 7) // 8 * 8
    buf = torch.full((b * n + 8, kp), 7.0, dtype=torch.bfloat16, device="cuda")
    out = buf[4:-4]
    col = 0
    for i, p in enumerate(parts):
        p2 = p.float().flatten(0, 1).cuda()
        ops.l2norm_pack(p2, out, col, True, zero_upto=kp if i == len(parts) - 1 else 0)
        col += p.shape[-1]
    torch.cuda.synchronize()
    g = buf.cpu()
    assert (g[:4] == 7).all() and (g[-4:] == 7).all(), "guard rows overwritten"
    o = g[4:-4]
    return hashlib.sha1(o.view(torch.int16).numpy().tobytes()).hexdigest()[:12], bool(torch.isfinite(o.float()).all()), float(o.float().abs().max())
for rep in (1, 5):
    obj = [bd["pad_obj_features"].repeat_interleave(rep, 0)]
    ocr = [bd[k].repeat_interleave(rep, 0) for k in ("ocr_fasttext", "ocr_phoc", "pad_ocr_features")]
    print("rep", rep, "obj", pack(obj, 0), "ocr", pack(ocr, 50))
