"""Potential of running two half-batches on two HIP streams: backbone forward (no_grad) B=8 on one stream vs 2 x B=4 concurrently."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd.host import modules, synth
fe = modules.resnet_feature_extractor("resnet101", freeze_bn=True, pretrained_backbone=False)
synth.load_formula_weights(fe)
fe = fe.cuda()
x = torch.from_numpy(synth.synth_image(8, 769, 769, seed=1)).cuda()
xa, xb = x[:4].contiguous(), x[4:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def one():
    with torch.no_grad():
        return fe(x)

def two():
    with torch.no_grad():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            a = fe(xa)
        with torch.cuda.stream(s2):
            b = fe(xb)
        cur.wait_stream(s1); cur.wait_stream(s2)
        return a, b

xs4 = [x[i:i + 2].contiguous() for i in range(0, 8, 2)]
st4 = [torch.cuda.Stream() for _ in range(4)]


def four():
    with torch.no_grad():
        cur = torch.cuda.current_stream()
        outs = []
        for st, xx in zip(st4, xs4):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(fe(xx))
        for st in st4:
            cur.wait_stream(st)
        return outs


for fn, name in ((one, "B=8 one stream"), (two, "2 x B=4 two streams"), (four, "4 x B=2 four streams"), (one, "B=8 one stream"), (two, "2 x B=4 two streams"), (four, "4 x B=2 four streams")):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print("%-22s %.2f ms per forward" % (name, (time.perf_counter() - t0) * 100))
