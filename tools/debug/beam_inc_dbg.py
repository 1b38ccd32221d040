"""debug: where do incremental and full beam steps disagree?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_decode_gpu import _models, _batch
from sam_textvqa_amd.registry import registry
from sam_textvqa_amd.synthetic import clone_batch
model, ref, shapes = _models(layers=("n", "s"))
eos, beam, s = 2, 5, shapes[3]
registry.EOS_IDX, registry.BOS_IDX = eos, 1
bd_cpu = _batch(6, shapes, 300, 31, "cpu")
bd_cpu["train_prev_inds"] = torch.zeros_like(bd_cpu["train_prev_inds"]); bd_cpu["train_prev_inds"][:, 0] = 1
bd_cpu["question_id"] = torch.arange(6) + 10
model.set_beam_size(beam)
outs = {}
for inc in ("0", "1"):
    os.environ["SAM_BEAM_INCREMENTAL"] = inc
    model.__dict__.pop("_sam_decode_sessions", None)
    bd = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in clone_batch(bd_cpu).items()}
    with torch.no_grad():
        got = model(bd, use_beam_search=True)
    outs[inc] = {k: got[k].float().cpu() for k in ("complete_seqs", "topkscores", "textvqa_scores")}
    outs[inc]["dec"] = bd["mmt_dec_output"].float().cpu()
a, b = outs["0"], outs["1"]
sa, sb = a["textvqa_scores"], b["textvqa_scores"]
seq = a["complete_seqs"].reshape(-1, s).long()
print("seqs equal:", bool((a["complete_seqs"] == b["complete_seqs"]).all()), "topk max diff", float((a["topkscores"] - b["topkscores"]).abs().max()))
for r in range(seq.shape[0]):
    d = (sa[r] - sb[r]).abs().amax(-1)          # per position
    dd = (a["dec"][r] - b["dec"][r]).abs().amax(-1)
    bad = (d > 0.05).nonzero().flatten().tolist()
    if bad:
        eos_pos = (seq[r] == eos).nonzero().flatten().tolist()
        print("row", r, "seq", seq[r].tolist(), "first eos at", eos_pos[:1], "score-diff positions", bad, "dec-diff", [round(float(x), 2) for x in dd])

# --- the fp32 oracle's search on the same weights: which mode matches its per-position scores?
from oracle import beam_search as OBS
with torch.no_grad():
    want, _, trace = OBS.forward_beam_search(ref, clone_batch(bd_cpu), beam, eos)
ws = want["textvqa_scores"].float()
print("oracle seqs equal ours:", bool((want["complete_seqs"].reshape(-1, s) == seq).all()))
for r in range(seq.shape[0]):
    da, db = (sa[r] - ws[r]).abs().amax(-1), (sb[r] - ws[r]).abs().amax(-1)
    if (da > 0.05).any() or (db > 0.05).any():
        print("row", r, "full-mode vs oracle per position", [round(float(x), 2) for x in da], "incremental vs oracle", [round(float(x), 2) for x in db])

print("---- detail")
for r in (14, 18, 19, 29, 13):
    am = lambda t: [int(t[r, p].argmax()) for p in range(8, 12)]
    mx = lambda t: [round(float(t[r, p].max()), 2) for p in range(8, 12)]
    print("row", r, "seq", seq[r, 8:].tolist(), "| argmax@8..11 full", am(sa), mx(sa), "inc", am(sb), mx(sb), "oracle", am(ws), mx(ws))
print("topk full", a["topkscores"].reshape(6, beam)[2:4].tolist()); print("topk oracle", want["topkscores"].reshape(6, beam)[2:4].tolist())
