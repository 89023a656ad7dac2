import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0,'tests')
from oracle import ref_pranet as rp
from rnd_semantic_segmentation_amd.host import synth
def _pn(name, shape, s=1.0): return (synth.uniform("pn." + name, shape) * s).astype(np.float32)
ds = torch.nn.Sequential(torch.nn.AvgPool2d(2, 2, ceil_mode=True, count_include_pad=False), torch.nn.Conv2d(64, 128, 1, 1, bias=False), torch.nn.BatchNorm2d(128))
cases=[("b2n_normal", rp.Bottle2neck(64, 16), [_pn("b2n_normal.x", (2, 64, 12, 12), 2)]),
 ("b2n_stage", rp.Bottle2neck(64, 32, stride=2, downsample=ds, stype="stage"), [_pn("b2n_stage.x", (2, 64, 13, 13), 2)]),
 ("rfb", rp.RFB(64, 32), [_pn("rfb.x", (2, 64, 11, 11), 2)]),
 ("agg", rp.PartialDecoder(32), [_pn("agg.x1", (2, 32, 3, 3)), _pn("agg.x2", (2, 32, 6, 6)), _pn("agg.x3", (2, 32, 12, 12))])]
def cos(a,b): return float(a.flatten().double()@b.flatten().double()/(a.double().norm()*b.double().norm()+1e-300))
for tag, mod, inputs in cases:
    synth.load_formula_weights(mod, prefix=tag+"."); mod.train()
    res={}
    for ac in (False, True):
        mod.zero_grad()
        xs=[torch.from_numpy(a).requires_grad_(True) for a in inputs]
        if ac:
            with torch.autocast('cpu', dtype=torch.bfloat16): y=mod(*xs)
        else: y=mod(*xs)
        y=y.float(); (y.square().mean()+y.mean()).backward()
        res[ac]=(y.detach(), [x.grad for x in xs], {k:p.grad.clone() for k,p in mod.named_parameters()})
    y0,dx0,g0=res[False]; y1,dx1,g1=res[True]
    gm=max(float(v.norm()) for v in g0.values())
    print(tag, 'out relmax %.2e'%((y1-y0).abs().max()/y0.abs().max()), 'dx relL2 %.2e'%max(float((a-b).norm()/b.norm()) for a,b in zip(dx1,dx0)),
          '|grad| %.2e'%max(abs(float(g1[k].norm()/g0[k].norm())-1) for k in g0 if float(g0[k].norm())>1e-3*gm),
          '1-cos %.2e'%(1-min(cos(g1[k],g0[k]) for k in g0 if float(g0[k].norm())>1e-3*gm)))
