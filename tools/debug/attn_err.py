import math, sys, torch, numpy as np
sys.path.insert(0, ".")
from tests.test_attention_gpu import make_problem, oracle_attention
from tests.util import unpack_bits
from oracle import sa_m4c_oracle as O
from sam_textvqa_amd import ops
B,H,hd=3,12,64
pr=make_problem(3,20,100,50,12,seed=2); N=pr["N"]; dev="cuda"
base=ops.mask_bits_prefix_lm(pr["key_valid"].to(torch.uint8).to(dev), pr["n_dec"])
bits=ops.mask_bits_spatial(base, pr["adj"].to(dev), pr["T"], H, (1,2))
allow=O.allow_mask(pr["key_valid"], pr["T"], pr["n_oo"], pr["n_dec"], pr["adj"], (1,2), H)
g=torch.Generator().manual_seed(5)
for std in (1.5, 0.5):
    qkv=(torch.randn(B*N,3*H*hd,generator=g)*std).to(torch.bfloat16)
    dout=torch.randn(B*N,H*hd,generator=g).to(torch.bfloat16)
    scale=1/8
    out,lse2,_=ops.attn_fwd(qkv.to(dev),bits,B,H,scale)
    dqkv=ops.attn_bwd(dout.to(dev),qkv.to(dev),lse2,bits,None,B,H,scale).float().cpu()
    x=qkv.float().requires_grad_(True)
    ro,_=oracle_attention(x,allow,B,H,scale); (ro*dout.float()).sum().backward()
    ref=x.grad
    # bf16-emulating oracle: P and dS rounded to bf16 before the second-stage matmuls
    xx=qkv.float().view(B,N,3,H,hd).permute(2,0,3,1,4); q,k,v=xx[0],xx[1],xx[2]
    s=(q@k.transpose(-1,-2))*scale; s=s.masked_fill(~allow,float("-inf"))
    alive=allow.any(-1,keepdim=True)
    p=torch.where(alive, torch.softmax(torch.where(alive, s, torch.zeros_like(s)),-1), torch.zeros_like(s)); p=torch.where(allow,p,torch.zeros_like(p))
    do=dout.float().view(B,N,H,hd).permute(0,2,1,3)
    dp=do@v.transpose(-1,-2); delta=(p*dp).sum(-1,keepdim=True)
    ds=(p*(dp-delta)*scale).bfloat16().float(); pb=p.bfloat16().float()
    edq=ds@k; edk=ds.transpose(-1,-2)@q; edv=pb.transpose(-1,-2)@do
    emu=torch.stack([edq,edk,edv],0).permute(1,3,0,2,4).reshape(B*N,3*H*hd)
    for nm,sl in (("dq",slice(0,768)),("dk",slice(768,1536)),("dv",slice(1536,2304))):
        r=ref[:,sl]; gq=dqkv[:,sl]; e=emu[:,sl]
        print("std %.1f %s: max|ref| %.3f  err(hip,ref) %.4f  err(emu,ref) %.4f  err(hip,emu) %.4f  rms(hip-ref) %.5f" % (std,nm,r.abs().max(),(gq-r).abs().max(),(e-r).abs().max(),(gq-e).abs().max(),(gq-r).pow(2).mean().sqrt()))
    print("fwd err", (out.float().cpu()-ro.detach()).abs().max().item(), ro.abs().max().item())
