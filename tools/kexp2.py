import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import kernels as K
from rnd_semantic_segmentation_amd.host import synth
from tools.kbench import timeit
B,h,w,Kc,H,W=8,97,97,19,769,769
low=(torch.randn((B,h,w,Kc))*1).cuda()
lab=torch.from_numpy(synth.synth_label(B,H,W,Kc,seed=9)).cuda().long()
lab_all=torch.full_like(lab,255)
print('loss+grad  %.1f us'%(timeit(lambda: K.upsample_ce(low,lab),20)*1e6))
print('loss only  %.1f us'%(timeit(lambda: K.upsample_ce(low,lab,want_grad=False),20)*1e6))
print('all ignored loss+grad %.1f us'%(timeit(lambda: K.upsample_ce(low,lab_all),20)*1e6))
print('inference tail %.1f us'%(timeit(lambda: K.upsample_softmax(low,(H,W)),20)*1e6))
