import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import ref_pranet as rp
from rnd_semantic_segmentation_amd.host import pranet, synth
import torch.nn.functional as F
net, ref = pranet.PraNet(), rp.PraNet()
synth.load_formula_weights(net, prefix="pranet."); synth.load_formula_weights(ref, prefix="pranet.")
net.cuda().train(); ref.train()
S=int(sys.argv[1]) if len(sys.argv)>1 else 96
x = torch.from_numpy(synth.synth_image(2, S, S, seed=31))
acts={}
def hook(name):
    def f(m,i,o): acts[name]=o.detach()
    return f
r=ref.resnet
for li in (1,2,3,4):
    for bi,b in enumerate(getattr(r,'layer%d'%li)): b.register_forward_hook(hook('resnet.layer%d.%d'%(li,bi)))
ref.rfb2_1.register_forward_hook(hook('rfb2')); ref.rfb3_1.register_forward_hook(hook('rfb3')); ref.rfb4_1.register_forward_hook(hook('rfb4'))
ref.agg1.register_forward_hook(hook('coarse'))
r.conv1[5].register_forward_hook(hook('stem1'))
with torch.no_grad():
    acts['stem']=r.stem(x)
    routs=ref(x)
net._taps={}
with torch.no_grad():
    outs=net(x.cuda())
for k,v in net._taps.items():
    a=v.t.float().permute(0,3,1,2).cpu(); b=acts[k]
    print('%-20s shape %-22s rel-max %.3e  rel-L2 %.3e  |ref|max %.3e' % (k, tuple(b.shape), (a-b).abs().max()/b.abs().max(), (a-b).norm()/b.norm(), b.abs().max()))
for i,(o,ro) in enumerate(zip(outs,routs)):
    a=o.float().cpu(); print('map',i,'rel-max %.3e rel-L2 %.3e' % ((a-ro).abs().max()/ro.abs().max(), (a-ro).norm()/ro.norm()))
