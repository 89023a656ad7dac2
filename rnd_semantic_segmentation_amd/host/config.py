"""Config surface of the reference (`from core.configs import cfg`), re-implemented without yacs.

Reference: core/configs/defaults.py:5-91 builds a yacs CfgNode; scripts call
cfg.merge_from_file(yaml) / cfg.merge_from_list([K, V, ...]) / cfg.freeze()
(train_src.py:58-60, test.py:67-69).  yacs is not installed in this image, so this is a
small clone with the behaviours the scripts rely on: attribute access, YAML merge with
literal_eval of string leaves ('5e-4' -> float, 'None' -> None), type checking with
list<->tuple coercion, unknown-key rejection, freeze.
"""
import copy
from ast import literal_eval

import yaml

_VALID = (tuple, list, str, int, float, bool, type(None))


class CfgNode(dict):
    IMMUTABLE = "__immutable__"

    def __init__(self, init=None):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # attribute access -------------------------------------------------------------------------
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError("Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        if name in self.__dict__:
            raise AttributeError("Invalid attempt to modify internal CfgNode state: {}".format(name))
        self[name] = value

    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def _set_frozen(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        out.__dict__[CfgNode.IMMUTABLE] = self.is_frozen()
        return out

    # merging ----------------------------------------------------------------------------------
    @staticmethod
    def _decode(v):
        """yacs semantics: strings go through literal_eval when they parse ('5e-4', 'None', '[1, 2]')."""
        if isinstance(v, dict):
            return v
        if not isinstance(v, str):
            return v
        try:
            return literal_eval(v)
        except (ValueError, SyntaxError):
            return v

    @staticmethod
    def _coerce(new, old, key):
        if old is None or new is None or type(new) is type(old):
            return new
        for a, b in ((list, tuple), (tuple, list)):
            if isinstance(new, a) and isinstance(old, b):
                return b(new)
        if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
            return float(new)
        raise ValueError("Type mismatch ({} vs. {}) with values ({} vs. {}) for config key: {}".format(
            type(old), type(new), old, new, key))

    def _merge(self, other, path):
        for k, v in other.items():
            full = ".".join(path + [k])
            if k not in self:
                raise KeyError("Non-existent config key: {}".format(full))
            v = self._decode(copy.deepcopy(v))
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError("Config key {} expects a mapping".format(full))
                self[k]._merge(v, path + [k])
            else:
                if not isinstance(v, _VALID):
                    raise ValueError("Key {} with value {} is not a valid type".format(full, type(v)))
                dict.__setitem__(self, k, self._coerce(v, self[k], full))

    def merge_from_file(self, path):
        if self.is_frozen():
            raise AttributeError("CfgNode is immutable")
        with open(path, "r") as f:
            loaded = yaml.safe_load(f) or {}
        self._merge(loaded, [])

    def merge_from_other_cfg(self, other):
        self._merge(other, [])

    def merge_from_list(self, opts):
        if self.is_frozen():
            raise AttributeError("CfgNode is immutable")
        opts = list(opts or [])
        if len(opts) % 2:
            raise AssertionError("Override list has odd length: {}; it must be a list of pairs".format(opts))
        for full, v in zip(opts[0::2], opts[1::2]):
            node = self
            parts = full.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError("Non-existent key: {}".format(full))
                node = node[p]
            leaf = parts[-1]
            if leaf not in node:
                raise KeyError("Non-existent key: {}".format(full))
            dict.__setitem__(node, leaf, self._coerce(self._decode(v), node[leaf], full))

    def __str__(self):
        def fmt(node, indent):
            lines = []
            for k in sorted(node):
                v = node[k]
                if isinstance(v, CfgNode):
                    lines.append(" " * indent + "{}:".format(k))
                    lines.extend(fmt(v, indent + 2))
                else:
                    lines.append(" " * indent + "{}: {}".format(k, v))
            return lines

        return "\n".join(fmt(self, 0))

    __repr__ = __str__


def default_tree():
    """Same keys / default values as reference core/configs/defaults.py:7-91 (the YAML surface)."""
    return {
        "MODEL": {"NAME": "deeplab_resnet101", "NUM_CLASSES": 2, "DEVICE": "cuda", "WEIGHTS": "", "FREEZE_BN": False},
        "INPUT": {
            "TRAINSIZE": 352, "SOURCE_INPUT_SIZE_TRAIN": (1280, 720), "TARGET_INPUT_SIZE_TRAIN": (1024, 512),
            "INPUT_SIZE_TEST": (1024, 512), "INPUT_SCALES_TRAIN": (1.0, 1.0), "IGNORE_LABEL": 255,
            "PIXEL_MEAN": [0.485, 0.456, 0.406], "PIXEL_STD": [0.229, 0.224, 0.225], "TO_BGR255": False,
            "BRIGHTNESS": 0.0, "CONTRAST": 0.0, "SATURATION": 0.0, "HUE": 0.0, "HORIZONTAL_FLIP_PROB_TRAIN": 0.0,
        },
        "AUG": {"NAME": "attn", "BLUR_PROB": 0.7, "ROTATE_PROB": 0.7, "JITTER_PROB": 0.7, "FLIP_PROB": 0.7, "PROB": 0.7,
                "COLLATE": "attn"},
        "DATASETS": {"DATASET_DIR": "", "SOURCE_TRAIN": "", "TARGET_TRAIN": "", "VALIDATION": "", "TEST": "", "CROSS_VAL": 0},
        "SOLVER": {
            "EPOCHS": 5, "MAX_ITER": 16000, "STOP_ITER": 10000, "LR_METHOD": "poly", "BASE_LR": 0.02, "BASE_LR_D": 0.008,
            "LR_POWER": 0.9, "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.0005, "WEIGHT_DECAY_BIAS": 0, "DECAY_RATE": 0.1,
            "DECAY_EPOCH": 50, "GAMMA": 0.1, "CHECKPOINT_PERIOD": 5, "BATCH_SIZE": 8, "BATCH_SIZE_VAL": 1,
        },
        "TEST": {"BATCH_SIZE": 1, "PRECISION": "fp32"},   # PRECISION (not in the reference): fp32 = exact evaluation path, bf16 = training engine
        "OUTPUT_DIR": ".",
        "resume": "",
        "PSEUDO_DIR": "",
    }


_C = CfgNode(default_tree())
cfg = _C   # `from core.configs import cfg` (reference core/configs/__init__.py:1)
