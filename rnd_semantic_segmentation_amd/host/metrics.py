"""Host-side utilities the DeepLab path uses from the reference's core/utils/utility.py and
core/utils/adapt_lr.py, re-stated (same names, argument meaning and outputs).

Differences that are deliberate and documented:
 * intersectionAndUnionGPU / confusion_matrix use integer bincount on the tensor's own device (the
   reference runs float histc on CPU and a per-pixel Python loop, utility.py:157-159, :347-359);
   the integers produced are identical (tests/test_host_logic.py pins them to the reference's outputs).
 * setup_logger creates OUTPUT_DIR (the reference crashes if it is missing, utility.py:243).
 * strip_prefix_if_present works when the prefix is present (the reference forgets to import
   OrderedDict, utility.py:167).
 * inference() runs the upsample + softmax tail in one HIP kernel (mi_upsample_softmax).
"""
import json
import logging
import os
from collections import OrderedDict, defaultdict, deque

import numpy as np
import torch


# ----------------------------------------------------------------------------- learning rate (adapt_lr.py:12-17)
def adjust_learning_rate(method, base_lr, iters, max_iter, power):
    if method == "poly":
        return base_lr * ((1 - float(iters) / max_iter) ** power)
    raise NotImplementedError(method)


# ----------------------------------------------------------------------------- meters (utility.py:24-131)
class AverageMeter(object):
    """Per-class intersection / union / target / output accumulators; macro = mean over images of per-image
    ratios, micro = ratio of sums (utility.py:24-72)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.intersection_sum = 0
        self.union_sum = 0
        self.target_sum = 0
        self.res_sum = 0
        self.count = 0
        self.iou_sum = 0
        self.f1_sum = 0

    def update(self, intersection, union, target, res):
        self.iou_sum = self.iou_sum + intersection / (union + 1e-10)
        self.f1_sum = self.f1_sum + 2 * intersection / (target + res + 1e-10)
        self.intersection_sum = self.intersection_sum + intersection
        self.union_sum = self.union_sum + union
        self.target_sum = self.target_sum + target
        self.res_sum = self.res_sum + res
        self.count += 1

    def results(self):
        n = float(self.count)
        macro_f1, macro_iou = self.f1_sum / n, self.iou_sum / n
        micro_f1 = 2 * self.intersection_sum / (self.target_sum + self.res_sum + 1e-10)
        micro_iou = self.intersection_sum / (self.union_sum + 1e-10)
        return dict(macro_iou=macro_iou, macro_f1=macro_f1, micro_iou=micro_iou, micro_f1=micro_f1)

    def summary(self, logger, num_classes=2):
        r = self.results()
        logger.info("Macro metric, val result: mIoU/mF1 {:.4f}/{:.4f}.".format(np.mean(r["macro_iou"]), np.mean(r["macro_f1"])))
        logger.info("Micro metric, val result: mIoU/mF1 {:.4f}/{:.4f}.".format(np.mean(r["micro_iou"]), np.mean(r["micro_f1"])))
        for i in range(num_classes):
            logger.info("Macro metric, class {} iou/f1 score: {:.4f}/{:.4f}.".format(i, r["macro_iou"][i], r["macro_f1"][i]))
            logger.info("Micro metric, class {} iou/f1 score: {:.4f}/{:.4f}.".format(i, r["micro_iou"][i], r["micro_f1"][i]))


class SmoothedValue(object):
    def __init__(self, window_size=20):
        self.deque = deque(maxlen=window_size)
        self.series = []
        self.total = 0.0
        self.count = 0

    def update(self, value):
        self.deque.append(value)
        self.series.append(value)
        self.count += 1
        self.total += value

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque)).mean().item()

    @property
    def global_avg(self):
        return self.total / self.count


class MetricLogger(object):
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                v = v.item()
            assert isinstance(v, (float, int))
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.__dict__.get("meters", {}):
            return self.meters[attr]
        raise AttributeError("'{}' object has no attribute '{}'".format(type(self).__name__, attr))

    def __str__(self):
        return self.delimiter.join("{}: {:.4f} ({:.4f})".format(n, m.median, m.global_avg) for n, m in self.meters.items())


# ----------------------------------------------------------------------------- segmentation metrics
def _hist(values, K):
    values = values[(values >= 0) & (values < K)]
    return torch.bincount(values, minlength=K)[:K]


def intersectionAndUnionGPU(output, target, K, ignore_index=255):
    """utility.py:148-161.  Returns float32 tensors (area_intersection, area_union, area_target, area_output)
    on the inputs' device; like the reference it overwrites `output` where target is ignored."""
    assert output.dim() in [1, 2, 3]
    assert output.shape == target.shape
    output = output.reshape(-1)
    target = target.reshape(-1)
    output[target == ignore_index] = ignore_index
    inter = output[output == target]
    ai = _hist(inter.long(), K).float()
    ao = _hist(output.long(), K).float()
    at = _hist(target.long(), K).float()
    return ai, ao + at - ai, at, ao


def intersectionAndUnion(output, target, K, ignore_index=255):
    """numpy twin, utility.py:133-145."""
    output = np.asarray(output).reshape(-1).copy()
    target = np.asarray(target).reshape(-1)
    output[target == ignore_index] = 255
    inter = output[output == target]
    ai, _ = np.histogram(inter, bins=np.arange(K + 1))
    ao, _ = np.histogram(output, bins=np.arange(K + 1))
    at, _ = np.histogram(target, bins=np.arange(K + 1))
    return ai, ao + at - ai, at, ao


def confusion_matrix(cfg, pd, gt):
    """utility.py:347-359: cmt[gt, pd] += 1 where gt != 255, int64 [K,K] on CPU."""
    K = cfg.MODEL.NUM_CLASSES
    pd = pd.reshape(-1).long()
    gt = gt.reshape(-1).long()
    keep = (gt != 255) & (gt >= 0) & (gt < K) & (pd >= 0) & (pd < K)
    return torch.bincount(gt[keep] * K + pd[keep], minlength=K * K).reshape(K, K).cpu()


def strip_prefix_if_present(state_dict, prefix):
    keys = sorted(state_dict.keys())
    if not all(key.startswith(prefix) for key in keys):
        return state_dict
    return OrderedDict((key.replace(prefix, ""), value) for key, value in state_dict.items())


def inference(feature_extractor, classifier, image, label, flip=True):
    """utility.py:179-191: 1/8-resolution logits -> bilinear(align_corners) to the LABEL size -> softmax,
    image 0 only ([1,K,H,W]); flip averages the horizontally mirrored pass."""
    size = tuple(label.shape[-2:])
    if flip:
        image = torch.cat([image, torch.flip(image, [3])], 0)
    with torch.no_grad():
        feat = feature_extractor(image)
        if hasattr(classifier, "predict_probs"):            # fused upsample + softmax kernel
            probs = classifier.predict_probs(feat, size)
        else:                                               # a substituted / foreign classifier: the reference's literal tail
            probs = torch.nn.functional.softmax(
                torch.nn.functional.interpolate(classifier(feat), size=size, mode="bilinear", align_corners=True), dim=1)
    if flip:
        out = (probs[0] + probs[1].flip(2)) / 2
    else:
        out = probs[0]
    return out.unsqueeze(dim=0)


# ----------------------------------------------------------------------------- io / logging
def soft_label_cross_entropy(pred, soft_label, pixel_weights=None):
    """utility.py:172-177 on materialised [N,C,H,W] tensors (API parity: AsppFada uses the fused kernel
    `PixelDiscriminator.soft_loss`, which never builds the full-resolution operands)."""
    loss = -soft_label.float() * torch.nn.functional.log_softmax(pred, dim=1)
    if pixel_weights is None:
        return torch.mean(torch.sum(loss, dim=1))
    return torch.mean(pixel_weights * torch.sum(loss, dim=1))


def load_json(path):
    with open(path, "r") as f:
        return json.load(f)


def dump_json(path, data):
    with open(path, "w") as f:
        json.dump(data, f)


def setup_logger(name, save_dir, distributed_rank=None):
    """utility.py:238-249 (same format string / handlers); creates save_dir, and only rank 0 writes the file."""
    os.makedirs(save_dir, exist_ok=True)
    logger = logging.getLogger(name)
    logger.setLevel(logging.INFO)
    logger.propagate = False
    if not logger.handlers:
        fmt = logging.Formatter("%(asctime)s [%(levelname)s] %(message)s")
        if not distributed_rank:
            fh = logging.FileHandler(os.path.join(save_dir, name + ".txt"))
            fh.setFormatter(fmt)
            logger.addHandler(fh)
        sh = logging.StreamHandler()
        sh.setFormatter(fmt)
        logger.addHandler(sh)
    return logger
