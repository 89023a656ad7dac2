"""ASPPTester: the reference's evaluation loop (core/testers/aspp_tester.py:10-83): inference() -> argmax ->
confusion matrix + IoU meters -> summary + aspp_confusion_matrix.json.  Device-agnostic host logic (the
reference hard-codes .cuda(), aspp_tester.py:57-58); metrics are integer bincounts on the device."""
import os

import numpy as np
import torch

from .metrics import (AverageMeter, confusion_matrix, dump_json, inference, intersectionAndUnionGPU,
                      strip_prefix_if_present)
from .modules import build_classifier, build_feature_extractor


class ASPPTester:
    build_feature_extractor = staticmethod(build_feature_extractor)
    build_classifier = staticmethod(build_classifier)

    def __init__(self, cfg, device, test_loader, logger, palette, trainid2name, saveres=False):
        self.cfg = cfg
        self.logger = logger
        self.test_loader = test_loader
        self.device = device
        self.palette = palette
        self.trainid2name = trainid2name
        self.saveres = saveres
        self.feature_extractor = self.build_feature_extractor(cfg)
        self.feature_extractor.to(device)
        self.classifier = self.build_classifier(cfg)
        self.classifier.to(device)
        # Evaluation reproduces the reference's fp32 masks (BASELINE: argmax identical, mIoU equal): TEST.PRECISION "fp32"
        # (default) selects the exact-fp32 schedule, "bf16" the 4x faster training engine.
        precision = cfg.TEST.PRECISION if "PRECISION" in cfg.TEST else "fp32"
        for m in (self.feature_extractor, self.classifier):
            if hasattr(m, "set_precision") and device.type == "cuda":
                m.set_precision(precision)

    def _load_checkpoint(self):
        self.logger.info("Loading checkpoint from {}".format(self.cfg.resume))
        checkpoint = torch.load(self.cfg.resume, map_location=self.device)
        self.feature_extractor.load_state_dict(strip_prefix_if_present(checkpoint["feature_extractor"], "module."))
        self.classifier.load_state_dict(strip_prefix_if_present(checkpoint["classifier"], "module."))

    def save_distill(self, output, name):
        """aspp_tester.py:33-45: palette PNG of the argmax mask under PSEUDO_DIR/inference/<dataset>."""
        from PIL import Image
        folder = os.path.join(self.cfg.PSEUDO_DIR, "inference", self.cfg.DATASETS.TEST)
        os.makedirs(folder, exist_ok=True)
        pred = output.cpu().numpy().squeeze().argmax(0).astype(np.uint8)
        mask = Image.fromarray(pred)
        mask.putpalette(list(self.palette))
        mask.save(os.path.join(folder, name[0] + ".png"))

    def test(self):
        num_classes = self.cfg.MODEL.NUM_CLASSES
        self.feature_extractor.eval()
        self.classifier.eval()
        self.meter = AverageMeter()
        cmt = torch.zeros(num_classes, num_classes, dtype=torch.int64)
        for x, y, name in self.test_loader:
            x = x.to(self.device, non_blocking=True)
            y = y.to(self.device, non_blocking=True).long()
            output = inference(self.feature_extractor, self.classifier, x, y, flip=False)     # [1,K,H,W]
            pred = output.max(1)[1]
            if self.saveres:
                self.save_distill(output, name)
            y0 = y[:1]                                    # inference() keeps image 0 only (utility.py:190)
            cmt = cmt + confusion_matrix(self.cfg, torch.flatten(pred), torch.flatten(y0))
            inter, union, target, res = intersectionAndUnionGPU(pred, y0, num_classes, self.cfg.INPUT.IGNORE_LABEL)
            self.meter.update(*[t.cpu().numpy() for t in (inter, union, target, res)])
        self.meter.summary(self.logger, num_classes)
        os.makedirs(self.cfg.OUTPUT_DIR, exist_ok=True)
        dump_json(os.path.join(self.cfg.OUTPUT_DIR, "aspp_confusion_matrix.json"),
                  {"cmt": cmt.tolist(), "classes": list(self.trainid2name.values())})
        return cmt
