"""Dataset surface: `build_dataset(cfg, mode, is_source)` / `build_collate_fn(cfg)` with the reference's
signatures (core/datasets/build.py:5-30), backed by a synthetic generator.

The reference's loaders read GTA5 / Cityscapes from disk through PIL/cv2/albumentations
(core/datasets/*.py, core/components/augment.py) - none of that data or those packages exist here, and
BASELINE.json's configs are defined on synthetic crops.  What is kept is the TENSOR CONTRACT of the loader
(core/datasets/transform.py:31-46, cityscapes.py:137-151, defaults.py:21-24):
    image  float32 [3,H,W], RGB/255 then (x-mean)/std   (here: unit-variance noise of that shape)
    label  float32 [H,W] with train-ids 0..K-1 and 255 = ignore
    name   str
"""
import os

import torch
from torch.utils.data import Dataset

from . import synth


class SyntheticSegmentation(Dataset):
    def __init__(self, cfg, mode="train", is_source=True, length=None):
        self.num_classes = cfg.MODEL.NUM_CLASSES
        if mode == "train":
            w, h = cfg.INPUT.SOURCE_INPUT_SIZE_TRAIN if is_source else cfg.INPUT.TARGET_INPUT_SIZE_TRAIN
        else:
            w, h = cfg.INPUT.INPUT_SIZE_TEST
        self.size = (int(h), int(w))
        env = os.environ.get("MI_SYNTH_LEN")
        self.length = int(length if length is not None else (env if env else (64 if mode == "train" else 4)))
        self.seed0 = {"train": 1000, "val": 2000, "test": 3000}[mode] + (0 if is_source else 500)

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        h, w = self.size
        img = synth.synth_image(1, h, w, seed=self.seed0 + i)[0]
        lab = synth.synth_label(1, h, w, self.num_classes, seed=self.seed0 + i)[0]
        return torch.from_numpy(img), torch.from_numpy(lab), "synthetic_%06d" % i


def build_collate_fn(cfg):
    """core/datasets/build.py:7-13: DeepLab YAMLs set AUG.COLLATE: None -> default collate."""
    if cfg.AUG.COLLATE in ("attn", "pranet"):
        raise NotImplementedError("AUG.COLLATE=%r belongs to the attn/pranet model families (out of scope)" % cfg.AUG.COLLATE)
    return None


def build_dataset(cfg, mode="train", is_source=True):
    assert mode in ["train", "val", "test"]
    return SyntheticSegmentation(cfg, mode, is_source)
