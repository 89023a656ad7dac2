"""nn.Module surface of the DeepLabV2 path, same class names / constructor arguments / state_dict keys as
the reference, computing through the MI355X engine (engine.py -> C-ABI kernels).

  FrozenBatchNorm2d          reference core/components/layers.py:5-23
  resnet_feature_extractor   reference core/models/feature_extractor.py:34-52 (+ core/components/resnet.py)
  ASPP_Classifier_V2         reference core/models/classifiers/aspp/classifier.py:6-32

These modules run on the GPU only: their forward raises if given CPU tensors (the CPU restatement lives in
oracle/, as test infrastructure).  Every convolution of the training step, the 7x7/s2 stem included (patch matrix +
GEMM), runs on the library behind the C-ABI; the trainable-BatchNorm stem uses torch's max_pool2d.
"""
import logging
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from . import arch, engine


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine parameters: four buffers, no eps (layers.py:18-20).
    Inside the backbone its arithmetic is fused into the producing GEMM's epilogue; called standalone it
    applies the same fold with torch ops on the tensor's device."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def fold(self):
        scale = self.weight * self.running_var.rsqrt()
        return scale, self.bias - self.running_mean * scale

    def forward(self, x):
        scale, shift = self.fold()
        return x * scale.reshape(1, -1, 1, 1).to(x.dtype) + shift.reshape(1, -1, 1, 1).to(x.dtype)


def _require_gpu(x, who):
    if not x.is_cuda:
        raise _lib.MiError("%s runs on the MI355X only (got a %s tensor); the CPU restatement of this path is "
                           "oracle/ref_model.py and is test infrastructure, not a fallback" % (who, x.device))
    _lib.lib()


class resnet_feature_extractor(nn.Module):
    """Dilated ResNet (output stride 8) returning the layer4 map.  state_dict keys: backbone.conv1.weight ...
    backbone.layer4.2.bn3.running_var (520 keys for resnet101 with FrozenBN)."""

    def __init__(self, backbone_name, pretrained_weights=None, aux=False, pretrained_backbone=True, freeze_bn=False,
                 layers=None):
        super().__init__()
        # feature_extractor.py:37-39: FrozenBatchNorm2d when MODEL.FREEZE_BN, torch.nn.BatchNorm2d otherwise (624 state keys / 312
        # parameter tensors for resnet101; batch statistics in train(), running statistics in eval())
        self.freeze_bn = bool(freeze_bn)
        if aux:
            raise NotImplementedError("aux (layer3) output is not used by any DeepLab config")
        if layers is None:
            if backbone_name not in arch.LAYERS:
                raise NotImplementedError("backbone %r (have: %s)" % (backbone_name, sorted(arch.LAYERS)))
            layers = arch.LAYERS[backbone_name]
        self.plan = arch.bottleneck_plan(layers)
        self.out_channels = self.plan[-1].cout
        self.backbone = arch.Holder()
        self._add_conv(arch.STEM)
        for blk in self.plan:
            for c in arch.block_convs(blk):
                self._add_conv(c)
        self._engine = engine.StageEngine(self, self.plan)
        self._fp32 = engine.Fp32Backbone(self, self.plan)
        self.precision = "bf16"
        self._store = None
        if pretrained_backbone and pretrained_weights:
            self._load_pretrained(pretrained_weights)

    def set_precision(self, precision):
        """"bf16": the training engine (bf16 operands, fp32 accumulate).  "fp32": the forward-only exact-fp32 schedule
        (csrc/igemm_f32.hip) that evaluation uses to reproduce the reference's masks (cfg TEST.PRECISION)."""
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision %r (bf16 | fp32)" % (precision,))
        self.precision = precision
        return self

    def _add_conv(self, c):
        node = arch.node_at(self.backbone, c.key)
        w = torch.empty(c.cout, c.cin, c.k, c.k)
        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")      # resnet.py:153-155
        node.weight = nn.Parameter(w)
        parent, leaf = c.bn.rsplit(".", 1) if "." in c.bn else ("", c.bn)
        holder = arch.node_at(self.backbone, parent) if parent else self.backbone
        holder.add_module(leaf, FrozenBatchNorm2d(c.cout) if self.freeze_bn else nn.BatchNorm2d(c.cout))

    def _load_pretrained(self, src):
        """MODEL.WEIGHTS (resnet.py:211-215 uses mmcv.load_checkpoint on a URL).  There is no network here:
        a URL is skipped with a warning, a local file is loaded non-strictly into `backbone.`."""
        log = logging.getLogger(__name__)
        if "://" in src:
            log.warning("MODEL.WEIGHTS=%s is a URL; offline build - keeping the initialised weights", src)
            return
        if not os.path.exists(src):
            raise FileNotFoundError(src)
        sd = torch.load(src, map_location="cpu")
        sd = sd.get("state_dict", sd)
        own = self.backbone.state_dict()
        picked = {k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}
        self.backbone.load_state_dict(picked, strict=False)
        log.info("loaded %d/%d backbone tensors from %s", len(picked), len(own), src)

    # ---- flat parameter storage (one buffer: fused SGD + bucketed all-reduce work on ranges) ----
    def engine_parameters(self):
        return [(k, p) for k, p in self.named_parameters()]

    def ensure_flat(self):
        dev = self.backbone.conv1.weight.device
        if self._store is None or not self._store.intact() or self._store.data.device != dev:
            self._store = engine.FlatStore(self.engine_parameters(), dev)
        return self._store

    def _stem_fold(self):
        bn = self.backbone.bn1
        sig = (engine.bn_versions(bn), bn.weight.data_ptr())
        if getattr(self, "_stem_sig", None) != sig:
            self._stem_scale, self._stem_shift = engine.bn_fold(bn)
            self._stem_sig = sig
        return self._stem_scale, self._stem_shift

    def sync_batchnorm(self, on=True):
        """train_distill.py:53 (SyncBatchNorm.convert_sync_batchnorm): with MODEL.FREEZE_BN=False every BatchNorm2d of the backbone
        exchanges its raw batch sums over torch.distributed's default group."""
        for m in self.backbone.modules():
            if isinstance(m, nn.BatchNorm2d):
                m._mi_sync = bool(on)
        return self

    def forward(self, x):
        _require_gpu(x, "resnet_feature_extractor")
        if self.precision == "fp32":
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                raise _lib.MiError("the fp32 path is forward-only (evaluation): call it under torch.no_grad(), or "
                                   "set_precision('bf16') to train")
            return self._fp32.forward(x.float()).permute(0, 3, 1, 2)        # NCHW-shaped view of NHWC fp32 memory
        self.ensure_flat()
        bb = self.backbone
        xb = x.to(dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        self._engine.batch_stats = (not self.freeze_bn) and self.training
        if self._engine.batch_stats:
            # trainable BatchNorm2d in train(): batch statistics between each conv and its normalise pass, two autograd nodes with hand-written
            # backward (engine.BnStemFn, engine.BnStagesFn)
            y = engine.BnStemFn.apply(xb, bb.conv1.weight, bb.bn1.weight, bb.bn1.bias, bb.bn1)
            params = [p for rt in self._engine.convs for p in (rt.weight, rt.bn.weight, rt.bn.bias)]
            return engine.BnStagesFn.apply(y, self._engine, *params).permute(0, 3, 1, 2)
        # stem: 7x7/2 conv as patch matrix + GEMM (engine.stem_conv_forward), then FrozenBN + ReLU + 3x3/2 max-pool in one HIP kernel
        scale, shift = self._stem_fold()
        y = engine.StemFn.apply(xb, bb.conv1.weight, scale, shift)                       # [B,Hp,Wp,64] bf16 NHWC
        weights = [rt.weight for rt in self._engine.convs]
        feat = engine.StagesFn.apply(y, self._engine, *weights)
        return feat.permute(0, 3, 1, 2)                  # NCHW-shaped view of NHWC memory (channels_last)


class ASPP_Classifier_V2(nn.Module):
    """DeepLabV2 ASPP head: sum of four dilated 3x3 convs (rate == padding), optional bilinear upsample.
    state_dict keys: conv2d_list.{0..3}.{weight,bias}; weights initialised N(0, 0.01) (classifier.py:23-24)."""

    def __init__(self, in_channels, dilation_series, padding_series, num_classes):
        super().__init__()
        if list(dilation_series) != list(padding_series) or len(dilation_series) != 4:
            raise NotImplementedError("ASPP engine expects 4 branches with padding == dilation (build.py:26-28 passes "
                                      "[6, 12, 18, 24] twice)")
        self.in_channels, self.num_classes = in_channels, num_classes
        self.rates = tuple(int(d) for d in dilation_series)
        self.conv2d_list = arch.Holder()
        # weights first, then biases: the four weight tensors are consecutive in the flat store
        for i in range(4):
            node = arch.Holder()
            node.weight = nn.Parameter(torch.randn(num_classes, in_channels, 3, 3) * 0.01)
            self.conv2d_list.add_module(str(i), node)
        bound = 1.0 / (in_channels * 9) ** 0.5
        for i in range(4):
            getattr(self.conv2d_list, str(i)).bias = nn.Parameter(torch.empty(num_classes).uniform_(-bound, bound))
        self._engine = engine.AsppEngine(self, self.rates, num_classes, in_channels)
        self._fp32 = engine.Fp32Aspp(self, self.rates)
        self.precision = "bf16"
        self._store = None

    def set_precision(self, precision):
        """See resnet_feature_extractor.set_precision; in "fp32" the head runs on the f32 MFMA for fp32 inputs (forward only)."""
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision %r (bf16 | fp32)" % (precision,))
        self.precision = precision
        return self

    def _low_fp32(self, x):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise _lib.MiError("the fp32 path is forward-only (evaluation): call it under torch.no_grad()")
        return self._fp32.forward(x.float().permute(0, 2, 3, 1).contiguous())

    def engine_parameters(self):
        named = dict(self.named_parameters())
        order = ["conv2d_list.%d.weight" % i for i in range(4)] + ["conv2d_list.%d.bias" % i for i in range(4)]
        return [(k, named[k]) for k in order]

    def ensure_flat(self):
        dev = self.conv2d_list._modules["0"].weight.device
        if self._store is None or not self._store.intact() or self._store.data.device != dev:
            self._store = engine.FlatStore(self.engine_parameters(), dev)
        return self._store

    def _params(self):
        return [p for _, p in self.engine_parameters()]

    @staticmethod
    def _nhwc(x):
        return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16) if x.dtype != torch.bfloat16 else x.permute(0, 2, 3, 1).contiguous()

    def forward(self, x, size=None):
        _require_gpu(x, "ASPP_Classifier_V2")
        if self.precision == "fp32":
            low = self._low_fp32(x)
            if size is not None:
                from .. import kernels
                return kernels.upsample_ac_fwd(low, tuple(int(s) for s in size))
            return low.permute(0, 3, 1, 2)
        self.ensure_flat()
        low = engine.AsppFn.apply(self._nhwc(x), self._engine, *self._params())      # [B,h,w,K] fp32
        if size is not None:
            return engine.UpsampleFn.apply(low, tuple(int(s) for s in size))           # [B,K,H,W] fp32
        return low.permute(0, 3, 1, 2)

    def loss(self, x, label, ignore_index=255, temperature=1.0):
        """criterion(self(x, label.shape[-2:]).div(temperature), label) fused (never writes the upsampled logits).
        Leaves the 1/8-resolution logits [B,K,h,w] (detached) in `self.last_low`."""
        _require_gpu(x, "ASPP_Classifier_V2")
        self.ensure_flat()
        out = engine.AsppLossFn.apply(self._nhwc(x), label.long().contiguous(), self._engine, int(ignore_index), float(temperature),
                                      *self._params())
        self.last_low = self._engine.last_low.detach().permute(0, 3, 1, 2)
        return out

    def predict_probs(self, x, size):
        """softmax(interpolate(self(x), size)) of reference utility.py:183-186 in one kernel; [B,K,H,W] fp32."""
        _require_gpu(x, "ASPP_Classifier_V2")
        from .. import kernels
        with torch.no_grad():
            if self.precision == "fp32":
                low = self._low_fp32(x)
            else:
                self._engine.prepare(False)
                low = self._engine.forward(self._nhwc(x))
            probs, _ = kernels.upsample_softmax(low, tuple(int(s) for s in size), want_pred=False)
        return probs


class CrossEntropyLoss(nn.Module):
    """torch.nn.CrossEntropyLoss(ignore_index=...) on the HIP kernel (aspp_trainer.py:61)."""

    def __init__(self, ignore_index=255):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, logits, target):
        _require_gpu(logits, "CrossEntropyLoss")
        return engine.SoftmaxCEFn.apply(logits.float(), target.long().contiguous(), self.ignore_index)


# ---- factories, same signatures as reference core/models/build.py:13-31 -----------------------------------------
def build_feature_extractor(cfg):
    model_name, backbone_name = cfg.MODEL.NAME.split("_")
    if backbone_name.startswith("resnet"):
        return resnet_feature_extractor(backbone_name, pretrained_weights=cfg.MODEL.WEIGHTS, aux=False,
                                        pretrained_backbone=True, freeze_bn=cfg.MODEL.FREEZE_BN)
    raise NotImplementedError("backbone %r: only the resnet family is on the MI355X hot path" % backbone_name)


def build_classifier(cfg):
    _, backbone_name = cfg.MODEL.NAME.split("_")
    if backbone_name.startswith("resnet"):
        return ASPP_Classifier_V2(2048, [6, 12, 18, 24], [6, 12, 18, 24], cfg.MODEL.NUM_CLASSES)
    raise NotImplementedError("backbone %r: only the resnet family is on the MI355X hot path" % backbone_name)


def build_adversarial_discriminator(cfg, num_features=None, mid_nc=256):
    from .fada import build_adversarial_discriminator as build
    return build(cfg, num_features, mid_nc)
