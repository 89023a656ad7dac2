"""Host run-ahead experiment: steps enqueued without a per-step sync; prints ms/step and the allocator's reserved memory.
usage: [MI_BATCH_LANES=1] [RUN_AHEAD=n] python tools/lanes_dbg.py"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, logging
from rnd_semantic_segmentation_amd.host import config as hc, synth
from rnd_semantic_segmentation_amd.host.trainer import ASPPTrainer
cfg = hc.CfgNode(hc.default_tree()); cfg.merge_from_file(os.path.join(bench.ROOT, "configs", "deeplabv2_r101_src.yaml")); cfg.freeze()
tr = ASPPTrainer("aspp", cfg, [None]*1000, 0, logger=logging.getLogger("x"))
if os.environ.get("RUN_AHEAD"):
    ASPPTrainer.RUN_AHEAD = int(os.environ["RUN_AHEAD"])
with torch.no_grad():
    for m in (tr.feature_extractor, tr.classifier):
        synth.load_formula_weights(m); m._store.generation += 1
x, lab = bench.synthetic_batch(8, 769, 0, torch.device("cuda"))
for i in range(4):
    tr.train_step(x, lab, 100000); tr.iteration += 1
torch.cuda.synchronize()
for rnd in range(2):
    t0 = time.perf_counter()
    for i in range(20):
        tr.train_step(x, lab, 100000); tr.iteration += 1
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    st = torch.cuda.memory_stats()
    print("run_ahead %d lanes %s: host %.1f ms/step total %.1f ms/step reserved %.2f GB segs %d retries %d" % (
        ASPPTrainer.RUN_AHEAD, os.environ.get("MI_BATCH_LANES", "2"), 1e3*(t1-t0)/20, 1e3*(t2-t0)/20, torch.cuda.memory_reserved()/2**30, st["segment.all.allocated"], st["num_alloc_retries"]))
