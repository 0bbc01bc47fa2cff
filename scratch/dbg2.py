import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, util_models as U
from finetune_fair_diffusion_amd import weights as W
from finetune_fair_diffusion_amd.classifier import MobileNetV3Large
from oracle import nn_mobilenet
dev=torch.device('cuda')
for gain, GS in ((1.0, 2.0**18), (1.2, 2.0**16), (1.3, 2.0**14)):
    sd=W.synthetic_state_dict(W.mobilenet_param_shapes(80), seed=4, gain=gain)
    sd={k:(v.half().float() if v.is_floating_point() else v) for k,v in sd.items()}
    mo=nn_mobilenet.MobileNetV3Large(80).eval(); mo.load_state_dict(sd); mo.requires_grad_(False)
    mp=MobileNetV3Large(sd, dev, 80)
    x=torch.randn(3,3,64,64,generator=torch.Generator().manual_seed(6)).clamp(-1,1)
    xr=x.half().float().requires_grad_(True)
    grads=[]
    h=xr
    feats=[]
    for f in mo.features:
        h=f(h); h.retain_grad(); feats.append(h)
    lo=mo.classifier(h.mean(dim=(2,3)))
    g=torch.zeros_like(lo); g[:,40]=0.3; g[:,41]=-0.3
    (lo*g).sum().backward()
    lp=mp.forward(x.to(dev).half(), record=True)
    tr=[]
    dch=mp.backward(g.to(dev), GS, trace=tr)
    print("gain",gain,"logits err", float((lp.float().cpu()-lo).abs().max()/lo.abs().max()))
    # tr[k] = grad wrt input of block (15-k) = output of features[15-k]  (scaled by 1024)
    for k,d in enumerate(tr):
        fi=15-k-1+1  # features index whose OUTPUT this is the grad of: block i (features[i+1]) input = features[i] output
        idx=15-k-1
        ref=feats[idx+0].grad  # features[idx] output
        B,C,H,Wd=ref.shape
        got=(d.float().cpu()/GS).reshape(B,H,Wd,C).permute(0,3,1,2)
        print(k, idx, tuple(ref.shape), float((got-ref).abs().max()/(ref.abs().max()+1e-30)), float(ref.abs().max()))
    print("dchips", float((dch.cpu()-xr.grad).abs().max()/xr.grad.abs().max()))
