import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import beam_search as OBS
from sam_textvqa_amd import ops
beam, vocab, n_ocr = 3, 300, 50
g = torch.Generator().manual_seed(100 + beam)
b, s, eos = 6, 7, 2
scores = [torch.randn(b * beam, vocab + n_ocr, generator=g) * 3 for _ in range(s)]
for t in range(1, s):
    scores[t][: 3 * beam, eos] += 6.0
scores[2][:, vocab + n_ocr - 2:] = -10000.0
dummy = {k: torch.zeros(b, 1) for k in OBS.BATCH_DICT_KEYS}
bd = dict(dummy, train_prev_inds=torch.zeros(b, s, dtype=torch.int64))
bd["train_prev_inds"][:, 0] = 1
obs = OBS.BeamSearch(beam, 1, eos)
bd = obs.init_batch(bd)
seqs = torch.zeros(b * beam, s, dtype=torch.int64).cuda(); seqs[:, 0] = 1
cum = torch.zeros(b * beam).cuda(); done = torch.zeros(b * beam, dtype=torch.uint8).cuda()
ctl = torch.zeros(4, dtype=torch.int32).cuda()
for t in range(s):
    full = torch.zeros(b * beam, s, vocab + n_ocr)
    full[:, t] = scores[t]
    bd["scores"] = full
    cum_before = bd["topkscores"].clone().float().reshape(-1)
    finish, bd, _ = obs.decode(bd, t)
    dev = full.cuda().view(b * beam * s, -1)
    ops.beam_step(dev[:, :vocab].contiguous(), dev[:, vocab:].contiguous(), b, beam, seqs, cum, done, eos, ctl=ctl)
    ok = torch.equal(seqs.cpu(), bd["train_prev_inds"])
    print("t", t, "ok", ok, "ctl", ctl.tolist(), "done", done.tolist())
    if not ok:
        bad = (seqs.cpu() != bd["train_prev_inds"]).any(1).nonzero().reshape(-1).tolist()
        for r in bad:
            print(" row", r, "gpu", seqs[r].tolist(), cum[r].item(), "ora", bd["train_prev_inds"][r].tolist(), bd["topkscores"][r].item())
        smp = bad[0] // beam
        print(" cum before:", cum_before[smp*beam:(smp+1)*beam].tolist())
        cs = torch.log(torch.sigmoid(scores[t][smp*beam:(smp+1)*beam]))
        print(" top logsig per beam:", cs.topk(4, -1))
        break
    if finish: break
