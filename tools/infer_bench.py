"""Evaluation-path timing (BASELINE configs[0] geometry: one 1024x512 image, flip-averaged like the reference's
core/utils/utility.py:179-191 `inference(..., flip=True)`): images/s of backbone + ASPP + upsample + softmax on one MI355X."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd.host import metrics, modules, synth
fe = modules.resnet_feature_extractor("resnet101", freeze_bn=True, pretrained_backbone=False)
cls = modules.ASPP_Classifier_V2(2048, [6, 12, 18, 24], [6, 12, 18, 24], 19)
synth.load_formula_weights(fe)
synth.load_formula_weights(cls)
fe, cls = fe.cuda().eval(), cls.cuda().eval()
for B in (1, 4):
    x = torch.from_numpy(synth.synth_image(B, 512, 1024, seed=3)).cuda()
    lab = torch.from_numpy(synth.synth_label(B, 512, 1024, 19, seed=3)).cuda()
    with torch.no_grad():
        for _ in range(3):
            out = metrics.inference(fe, cls, x, lab, flip=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            out = metrics.inference(fe, cls, x, lab, flip=True)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("batch %d, 512x1024, flip-averaged: %.2f ms per batch, %.1f images/s" % (B, dt * 1e3, B / dt))
