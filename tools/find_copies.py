"""Which host lines issue the device-to-device copies (`__amd_rocclr_copyBuffer`) of a training step: torch.profiler with stacks."""
import os, sys, collections, logging, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rnd_semantic_segmentation_amd.host import config as hc, synth
from rnd_semantic_segmentation_amd.host.trainer import ASPPTrainer
cfg = hc.CfgNode(hc.default_tree()); cfg.merge_from_file(os.path.join(bench.ROOT, "configs", "deeplabv2_r101_src.yaml")); cfg.freeze()
tr = ASPPTrainer("aspp", cfg, [None] * 1000, 0, logger=logging.getLogger("x"))
with torch.no_grad():
    for m in (tr.feature_extractor, tr.classifier):
        synth.load_formula_weights(m); m._store.generation += 1
x, lab = bench.synthetic_batch(8, 769, 0, torch.device("cuda"))
for _ in range(3):
    tr.train_step(x, lab, 100000); tr.iteration += 1
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(x, lab, 100000); tr.iteration += 1
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    n = ev.name.lower()
    if "memcpy" in n or "copy_" in n or "copybuffer" in n:
        st = [s for s in (ev.stack or []) if "rnd_semantic" in s or "bench" in s]
        cnt[(ev.name[:40], st[0][-90:] if st else "?")] += 1
for (name, where), c in cnt.most_common(25):
    print("%4d  %-40s %s" % (c, name, where))
