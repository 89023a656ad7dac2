import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import kernels as K, _lib
from tools.kbench import timeit
B,H=8,97
def run(ci,co,k,d,flags_extra,label):
    x=torch.randn((B,H,H,ci),device='cuda').to(torch.bfloat16)
    w=torch.randn((co,ci,k,k),device='cuda')*0.05
    wp=K.pack_weight_fwd(w); out=torch.empty((B,H,H,co),device='cuda',dtype=torch.bfloat16)
    sc=torch.rand(co,device='cuda')+0.5; sh=torch.randn(co,device='cuda')
    pad=d if k==3 else 0
    L=_lib.lib(); st=ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P=lambda t: ctypes.c_void_p(t.data_ptr())
    for fl,name,z in ((1|4,'bn+relu',0),(1|4|(1<<29),'stagger1',1),(1|4|(1<<29),'stagger2',2),(1|4|(1<<29),'stagger4',4),(1<<30,'nostore',0)):
        f=lambda: L.mi_conv_gemm(P(x),P(wp),P(out),B,H,H,ci,H,H,co,k,1,pad,d,0,P(sc),P(sh),None,None,fl,z,st)
        t=timeit(f,30); print('%-18s %-8s %7.1f us  %6.0f TF'%(label,name,t*1e6,2.0*B*H*H*ci*co*k*k/t/1e12))
run(256,1024,1,1,0,'1x1 256->1024')
run(1024,256,1,1,0,'1x1 1024->256')
run(256,256,3,2,0,'3x3 256 d2')
