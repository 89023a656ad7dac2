import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import ref_pranet as rp
from rnd_semantic_segmentation_amd.host import synth
S=int(sys.argv[1]) if len(sys.argv)>1 else 96
ref = rp.PraNet(); synth.load_formula_weights(ref, prefix="pranet."); ref.train()
x = torch.from_numpy(synth.synth_image(2, S, S, seed=31))
acts={}
cur=[None]
def hook(name):
    def f(m,i,o): acts.setdefault(cur[0],{})[name]=o.detach().float()
    return f
r=ref.resnet
for li in (1,2,3,4):
    for bi,b in enumerate(getattr(r,'layer%d'%li)): b.register_forward_hook(hook('layer%d.%d'%(li,bi)))
ref.rfb4_1.register_forward_hook(hook('rfb4')); ref.agg1.register_forward_hook(hook('coarse'))
with torch.no_grad():
    cur[0]='fp32'; o32=ref(x)
    cur[0]='bf16'
    with torch.autocast('cpu', dtype=torch.bfloat16):
        o16=ref(x)
for k in acts['fp32']:
    a,b=acts['bf16'][k],acts['fp32'][k]
    print('%-12s rel-L2 %.3e' % (k, (a-b).norm()/b.norm()))
for i in range(4):
    print('map',i,'rel-L2 %.3e' % ((o16[i].float()-o32[i]).norm()/o32[i].norm()))
