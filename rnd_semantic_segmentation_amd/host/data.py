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


class SyntheticPolyp(Dataset):
    """The polyp_train / polyp_val contract the PraNet trainer and tester consume (pranet_trainer.py:39-43, pranet_tester.py:27-34):
    image float32 [3,S,S] mean/std normalised, mask float32 [1,S,S] with values in {0, 1} (one or two smooth blobs), name.
    S = INPUT.TRAINSIZE in train mode (the reference's pra_trans resizes to it, augment.py:55-86), INPUT.INPUT_SIZE_TEST otherwise."""

    def __init__(self, cfg, mode="train", length=None):
        if mode == "train":
            self.size = (int(cfg.INPUT.TRAINSIZE), int(cfg.INPUT.TRAINSIZE))
        else:
            w, h = cfg.INPUT.INPUT_SIZE_TEST
            self.size = (int(h), int(w))
        env = os.environ.get("MI_SYNTH_LEN")
        self.length = int(length if length is not None else (env if env else (64 if mode == "train" else 4)))
        self.seed0 = {"train": 5000, "val": 6000, "test": 7000}[mode]

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        h, w = self.size
        img, mask = synth.synth_polyp(1, h, w, seed=self.seed0 + i)
        return torch.from_numpy(img[0]), torch.from_numpy(mask[0]), "synthetic_polyp_%06d" % i


def build_collate_fn(cfg):
    """core/datasets/build.py:5-13.  The reference's collate functions (core/datasets/func.py: attn_collate_fn, the missing pranet_collate_fn)
    turn numpy HWC images from disk into tensors; the synthetic datasets here already yield the tensors of the loader contract, so the
    default collation applies for every value of AUG.COLLATE ("attn" is the default of defaults.py and what the PraNet YAML inherits)."""
    return None


def build_dataset(cfg, mode="train", is_source=True):
    assert mode in ["train", "val", "test"]
    name = cfg.DATASETS.SOURCE_TRAIN if (mode == "train" and is_source) else (cfg.DATASETS.TARGET_TRAIN if mode == "train" else cfg.DATASETS.TEST)
    if "polyp" in str(name) or "kvasir" in str(name):           # dataset_path_catalog.py:36-51,99-106
        return SyntheticPolyp(cfg, mode)
    return SyntheticSegmentation(cfg, mode, is_source)
