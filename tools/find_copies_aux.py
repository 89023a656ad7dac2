"""Which host lines issue the device copies / fills of a PraNet or GALD training step (torch.profiler with stacks).
usage: python tools/find_copies_aux.py pranet|gald"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd.host import gald, pranet, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "pranet"
dev = torch.device("cuda")
torch.manual_seed(0)
if wl == "pranet":
    net = pranet.PraNet().to(dev).train()
    net.ensure_flat()
    opt = pranet.FlatAdam(net, 1e-4 / 8, grad_clamp=0.5)
    img, mask = synth.synth_polyp(16, 352, 352, seed=3)
    x, gt = torch.from_numpy(img).to(dev), torch.from_numpy(mask).to(dev)

    def step():
        opt.zero_grad()
        ls = [pranet.structure_loss(o, gt) for o in net(x)]
        (ls[3] + ls[2] + ls[1] + ls[0]).backward()
        opt.step()
else:
    enc, dec = gald.GCPAEncoder().to(dev).train(), gald.GCPADecoder().to(dev).train()
    enc.ensure_flat()
    dec.ensure_flat()
    oe, od = pranet.FlatAdam(enc, 1e-4), pranet.FlatAdam(dec, 1e-3)
    x = torch.from_numpy(synth.synth_image(6, 720, 1280, seed=9)).to(dev)
    lab = torch.from_numpy(synth.synth_label(6, 720, 1280, 19, seed=9)).to(dev).long()

    def step():
        oe.zero_grad()
        od.zero_grad()
        l5, l4, l3, l2 = dec.losses(x, enc(x), lab)
        (l2 * 1 + l3 * 0.8 + l4 * 0.6 + l5 * 0.4).backward()
        oe.step()
        od.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    n = ev.name.lower()
    if "memcpy" in n or "copy_" in n or "copybuffer" in n or "fill" in n or "zero_" in n or "elementwise" in n:
        st = [s for s in (ev.stack or []) if "rnd_semantic" in s or "find_copies" in s]
        key = (ev.name[:44], " <- ".join(s[-60:] for s in st[:2]) if st else "?")
        cnt[key] += 1
        dur[key] += ev.device_time_total if hasattr(ev, "device_time_total") else 0
for key, c in cnt.most_common(40):
    print("%4d %8.1f us  %-44s %s" % (c, dur[key], key[0], key[1]))
