"""Per-workgroup timeline of one conv launch (debug flag 1<<29 of the experiment build)."""
import os, sys, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import kernels as K, _lib
B, H = 8, 97
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ci, co = int(sys.argv[1]), int(sys.argv[2])
k = int(sys.argv[3]) if len(sys.argv) > 3 else 1
res = co > ci
x = torch.randn((B, H, H, ci), device='cuda').to(torch.bfloat16)
w = torch.randn((co, ci, k, k), device='cuda') * 0.05
wp = K.pack_weight_fwd(w)
out = torch.empty((B, H, H, co), device='cuda', dtype=torch.bfloat16)
r = torch.randn((B, H, H, co), device='cuda').to(torch.bfloat16)
sc = torch.rand(co, device='cuda') + 0.5
sh = torch.randn(co, device='cuda')
dbg = torch.zeros((8192, 8), dtype=torch.int64, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
fl = 1 | 4 | (2 if res else 0) | (1 << 29)
pad = 2 if k == 3 else 0
for it in range(3):
    dbg.zero_()
    L.mi_conv_gemm(P(x), P(wp), P(out), B, H, H, ci, H, H, co, k, 1, pad, 2 if k == 3 else 1, 0, P(sc), P(sh), P(r), P(dbg), None, fl, 0, ctypes.c_float(0.0), st)
    torch.cuda.synchronize()
d = dbg.cpu().numpy()
d = d[d[:, 0] != 0]
t0 = d[:, 0].min()
T = (d[:, :5] - t0).astype(np.float64)
hw = d[:, 5]
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh_ = (hw >> 12) & 1
print("workgroups", len(d), "span (ticks)", T[:, 4].max(), " tick assumed 10 ns (100 MHz) =>", T[:, 4].max() / 100.0, "us")
dur = T[:, 4] - T[:, 0]
print("per-WG ticks: total mean %.0f | prologue %.0f | main loop %.0f | epilogue issue %.0f | store drain %.0f" % (
    dur.mean(), (T[:, 1] - T[:, 0]).mean(), (T[:, 2] - T[:, 1]).mean(), (T[:, 3] - T[:, 2]).mean(), (T[:, 4] - T[:, 3]).mean()))
order = np.argsort(T[:, 0])
print("start-time histogram (ticks): ", np.histogram(T[:, 0], bins=16)[0])
print("first 12 WG rows [start, +prologue, +loop, +epi, +drain]:")
for i in order[:6].tolist() + order[600:606].tolist():
    print("  blk %5d cu %2d se %d  start %7.0f  %6.0f %6.0f %6.0f %6.0f" % (i, cu[i], se[i], T[i, 0], T[i, 1] - T[i, 0], T[i, 2] - T[i, 1], T[i, 3] - T[i, 2], T[i, 4] - T[i, 3]))
