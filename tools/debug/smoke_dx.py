import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import sa_m4c_oracle as O
from oracle import spatial_graph as SG
import sam_textvqa_amd.modules as M
from sam_textvqa_amd.synthetic import mmt_config_dict
torch.manual_seed(0)
T, n_obj, n_ocr, n_dec, B = 5, 20, 15, 5, 2
cfgd = mmt_config_dict(3, ("s",), n_dec=n_dec, T=T, n_obj=n_obj, n_ocr=n_ocr)
cfgd.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
o_layer = O.SpatialBertLayer(O.BertConfig.from_dict(cfgd)).eval()
with torch.no_grad():
    for p in o_layer.parameters():
        p.copy_((p if p.dim() == 1 else p.to(torch.bfloat16).float()))
layer = M.SpatialBertLayer(M.BertConfig.from_dict(cfgd)).eval()
layer.load_state_dict(o_layer.state_dict()); layer.cuda()
n = T + n_obj + n_ocr + n_dec
rng = np.random.RandomState(0)
boxes = rng.rand(B, n_obj + n_ocr, 2) * 0.8
boxes = np.concatenate([boxes, boxes + 0.02 + rng.rand(B, n_obj + n_ocr, 2) * 0.15], -1); boxes[1, -4:] = 0
adj = torch.from_numpy(np.stack([SG.compose(SG.relation_codes(b), 3) for b in boxes]))
qm, om, cm = torch.ones(B, T, dtype=torch.long), torch.ones(B, n_obj, dtype=torch.long), torch.ones(B, n_ocr, dtype=torch.long)
qm[0, 3:] = 0; cm[1, -4:] = 0
ext = O.MMT.extended_attention_mask(qm, om, cm, n_dec)
x = torch.randn(B, n, 768).to(torch.bfloat16)
xo = x.float().requires_grad_(True); yo = o_layer(xo, ext, adj)[0]; yo.square().sum().backward()
xg = x.cuda().requires_grad_(True); yg = layer(xg, ext.cuda(), adj.cuda())[0]; yg.float().square().sum().backward()
err = (xg.grad.float().cpu() - xo.grad).abs().amax(-1)
print("max |dx ref|", xo.grad.abs().max().item(), "max |dx hip|", xg.grad.float().abs().max().item())
print("row err b0", [round(v, 3) for v in err[0].tolist()])
print("row err b1", [round(v, 3) for v in err[1].tolist()])
for pn, p in layer.named_parameters():
    po = dict(o_layer.named_parameters())[pn]
    print(pn, "rel err", ((p.grad.cpu() - po.grad).abs().max() / (po.grad.abs().max() + 1e-12)).item())
