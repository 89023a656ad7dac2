"""Per-step durations of the first 60 DeepLab training steps of a fresh process (events after every step, no host sync in between): how many
steps the default bench needs before it runs at its sustained rate."""
import logging, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from rnd_semantic_segmentation_amd.host import config as hc, synth
from rnd_semantic_segmentation_amd.host.trainer import ASPPTrainer
cfg = hc.CfgNode(hc.default_tree()); cfg.merge_from_file(os.path.join(bench.ROOT, "configs", "deeplabv2_r101_src.yaml")); cfg.freeze()
tr = ASPPTrainer("aspp", cfg, [None] * 1000, 0, logger=logging.getLogger("x"))
with torch.no_grad():
    for m in (tr.feature_extractor, tr.classifier):
        synth.load_formula_weights(m); m._store.generation += 1
x, lab = bench.synthetic_batch(8, 769, 0, torch.device("cuda"))
evs = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
evs[0].record()
for i in range(60):
    tr.train_step(x, lab, 100000); tr.iteration += 1
    evs[i + 1].record()
torch.cuda.synchronize()
ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(60)]
print(" ".join("%.1f" % t for t in ts))
print("reserved %.1f GB" % (torch.cuda.memory_reserved() / 2**30))
