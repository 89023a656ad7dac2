import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import ref_pranet as rp
from rnd_semantic_segmentation_amd.host import pranet, synth
tag='b2n_normal'
x0=(synth.uniform("pn.b2n_normal.x",(2,64,12,12))*2).astype(np.float32)
mod, ref = pranet.Bottle2neck(64,16), rp.Bottle2neck(64,16)
synth.load_formula_weights(mod, prefix=tag+'.'); synth.load_formula_weights(ref, prefix=tag+'.')
mod.cuda().train(); ref.train()
x=torch.from_numpy(x0).cuda().requires_grad_(True)
y=mod(x); yf=y.float(); (yf.square().mean()+yf.mean()).backward()
rx=torch.from_numpy(x0).requires_grad_(True)
ry=ref(rx); (ry.square().mean()+ry.mean()).backward()
d=(x.grad.cpu()-rx.grad)
print('max err', d.abs().max().item(), 'ref max', rx.grad.abs().max().item(), 'ref mean abs', rx.grad.abs().mean().item())
print('err per channel (max):', d.abs().amax((0,2,3))[:16])
print('mean abs err', d.abs().mean().item())
# which fraction of elements is off by more than 2%
bad=(d.abs()>0.02*rx.grad.abs().max())
print('bad frac', bad.float().mean().item())
idx=bad.nonzero()[:10]
for i in idx:
    i=tuple(i.tolist()); print(i, x.grad.cpu()[i].item(), rx.grad[i].item(), 'out', ry[i].item(), y.float().cpu()[i].item())
