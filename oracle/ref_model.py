"""ORACLE (test infrastructure): stock-PyTorch CPU restatement of the reference's
DeepLabV2-ResNet + ASPP graph, with state_dict keys identical to the reference's
(`backbone.conv1.weight ... backbone.layer4.2.bn3.running_var`,
`conv2d_list.{0..3}.{weight,bias}`), written table-driven rather than
class-per-layer.  It is floating-point work, so a torch fp32/fp64 reference is the
permitted form of the oracle here; it is pinned to the reference by
tests/golden/tinynet_*.npz, r101_*.npz (made by oracle/make_golden.py from the
reference's own modules).

Restates:
  reference core/components/resnet.py:73-113 (Bottleneck), :118-191 (ResNet stem,
  _make_layer with stride->dilation), core/components/layers.py:5-23 (FrozenBN),
  core/models/feature_extractor.py:34-52 (layer4 output, 'backbone.' prefix),
  core/models/classifiers/aspp/classifier.py:6-32 (ASPP head),
  core/trainers/aspp_trainer.py:77-97 (one training step).

Also serves as bench.py's `cpu_baseline` ("port") and as the CPU stand-in model
for host-logic tests.  Never imported by the product package.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Box(nn.Module):
    """Bare container so parameter paths nest exactly like the reference's."""


def _box_path(root, path):
    node = root
    for part in path.split("."):
        if not hasattr(node, part):
            node.add_module(part, _Box())
        node = getattr(node, part)
    return node


def resnet_plan(layers=(3, 4, 23, 3), dilate=(False, True, True)):
    """Block table equivalent to reference resnet.py:_make_layer (169-191) calls at
    :142-148 with replace_stride_with_dilation=[False, True, True]
    (feature_extractor.py:42).  Yields dicts per bottleneck."""
    plan = []
    inplanes, dilation = 64, 1
    for li, (planes, nblk, stride) in enumerate(zip((64, 128, 256, 512), layers, (1, 2, 2, 2))):
        prev_dil = dilation
        if li > 0 and dilate[li - 1]:
            dilation *= stride
            stride = 1
        for b in range(nblk):
            first = b == 0
            plan.append(dict(
                name="layer%d.%d" % (li + 1, b), cin=inplanes, width=planes, cout=planes * 4,
                stride=stride if first else 1, dil=prev_dil if first else dilation,
                down=first and (stride != 1 or inplanes != planes * 4)))
            inplanes = planes * 4
    return plan


class RefFeatureExtractor(nn.Module):
    def __init__(self, layers=(3, 4, 23, 3), freeze_bn=True):
        super().__init__()
        self.freeze_bn = freeze_bn            # False: nn.BatchNorm2d semantics (feature_extractor.py:37-39), batch statistics in train()
        self.plan = resnet_plan(layers)
        self.backbone = _Box()
        self._conv("conv1", 64, 3, 7)
        self._bn("bn1", 64)
        for blk in self.plan:
            n = blk["name"]
            self._conv(n + ".conv1", blk["width"], blk["cin"], 1)
            self._bn(n + ".bn1", blk["width"])
            self._conv(n + ".conv2", blk["width"], blk["width"], 3)
            self._bn(n + ".bn2", blk["width"])
            self._conv(n + ".conv3", blk["cout"], blk["width"], 1)
            self._bn(n + ".bn3", blk["cout"])
            if blk["down"]:
                self._conv(n + ".downsample.0", blk["cout"], blk["cin"], 1)
                self._bn(n + ".downsample.1", blk["cout"])

    def _conv(self, path, o, c, k):
        box = _box_path(self.backbone, path)
        w = torch.empty(o, c, k, k)
        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")  # resnet.py:153-155
        box.weight = nn.Parameter(w)

    def _bn(self, path, n):
        box = _box_path(self.backbone, path)
        if self.freeze_bn:
            box.register_buffer("weight", torch.ones(n))
            box.register_buffer("bias", torch.zeros(n))
        else:                                  # torch.nn.BatchNorm2d(n): affine parameters, eps 1e-5, momentum 0.1
            box.weight = nn.Parameter(torch.ones(n))
            box.bias = nn.Parameter(torch.zeros(n))
        box.register_buffer("running_mean", torch.zeros(n))
        box.register_buffer("running_var", torch.ones(n))
        if not self.freeze_bn:
            box.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def _w(self, path):
        return _box_path(self.backbone, path).weight

    def _apply_bn(self, x, path):
        b = _box_path(self.backbone, path)
        if not self.freeze_bn:
            if self.training:
                b.num_batches_tracked += 1
            return F.batch_norm(x, b.running_mean, b.running_var, b.weight, b.bias, self.training, 0.1, 1e-5)
        scale = b.weight * b.running_var.rsqrt()            # layers.py:19 (no eps)
        shift = b.bias - b.running_mean * scale              # layers.py:20
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)

    def forward(self, x):
        x = F.conv2d(x, self._w("conv1"), None, 2, 3)
        x = F.relu(self._apply_bn(x, "bn1"))
        x = F.max_pool2d(x, 3, 2, 1)
        for blk in self.plan:
            n = blk["name"]
            idt = x
            y = F.relu(self._apply_bn(F.conv2d(x, self._w(n + ".conv1")), n + ".bn1"))
            y = F.conv2d(y, self._w(n + ".conv2"), None, blk["stride"], blk["dil"], blk["dil"])
            y = F.relu(self._apply_bn(y, n + ".bn2"))
            y = self._apply_bn(F.conv2d(y, self._w(n + ".conv3")), n + ".bn3")
            if blk["down"]:
                idt = self._apply_bn(F.conv2d(x, self._w(n + ".downsample.0"), None, blk["stride"]),
                                     n + ".downsample.1")
            x = F.relu(y + idt)
        return x


class RefASPP(nn.Module):
    def __init__(self, in_channels=2048, rates=(6, 12, 18, 24), num_classes=19):
        super().__init__()
        self.rates = tuple(rates)
        self.conv2d_list = _Box()
        for i in range(len(rates)):
            box = _Box()
            box.weight = nn.Parameter(torch.randn(num_classes, in_channels, 3, 3) * 0.01)  # classifier.py:23-24
            bound = 1.0 / math.sqrt(in_channels * 9)
            box.bias = nn.Parameter(torch.empty(num_classes).uniform_(-bound, bound))
            self.conv2d_list.add_module(str(i), box)

    def forward(self, x, size=None):
        out = None
        for i, r in enumerate(self.rates):
            box = getattr(self.conv2d_list, str(i))
            y = F.conv2d(x, box.weight, box.bias, 1, r, r)
            out = y if out is None else out + y
        if size is not None:
            out = F.interpolate(out, size=size, mode="bilinear", align_corners=True)
        return out


def ref_inference(fe, cls, image, label):
    """reference core/utils/utility.py:179-191, flip=False."""
    with torch.no_grad():
        out = cls(fe(image))
    out = F.interpolate(out, size=label.shape[-2:], mode="bilinear", align_corners=True)
    return F.softmax(out, dim=1)[0].unsqueeze(0)


def ref_train_step(fe, cls, opt_fea, opt_cls, image, label, it, max_iter, base_lr, power=0.9):
    """reference core/trainers/aspp_trainer.py:77-97 restated: poly LR (classifier x10),
    zero_grad, forward with upsample to label size, CE(ignore 255), backward, two SGD steps."""
    lr = base_lr * ((1 - float(it) / max_iter) ** power)
    for g in opt_fea.param_groups:
        g["lr"] = lr
    for g in opt_cls.param_groups:
        g["lr"] = lr * 10
    opt_fea.zero_grad()
    opt_cls.zero_grad()
    label = label.long()
    out = cls(fe(image), label.shape[-2:])
    loss = F.cross_entropy(out, label, ignore_index=255)
    loss.backward()
    opt_fea.step()
    opt_cls.step()
    return loss.detach(), lr


def make_optimizers(fe, cls, base_lr, momentum=0.9, weight_decay=5e-4):
    """reference aspp_trainer.py:25-26."""
    return (torch.optim.SGD(fe.parameters(), lr=base_lr, momentum=momentum, weight_decay=weight_decay),
            torch.optim.SGD(cls.parameters(), lr=base_lr * 10, momentum=momentum, weight_decay=weight_decay))


# ------------------------------------------------------------------------------------------------ FADA
class RefPixelDiscriminator(nn.Module):
    """reference core/models/discriminator.py:31-50: D = Conv3x3(C->ndf) LeakyReLU(.2) Conv3x3(ndf->ndf/2) LeakyReLU(.2);
    cls1, cls2 = Conv3x3(ndf/2 -> K); output cat(cls1, cls2) [B,2K,h,w], optional bilinear(align_corners) upsample.
    state_dict keys: D.0.*, D.2.*, cls1.*, cls2.*"""

    def __init__(self, input_nc=2048, ndf=256, num_classes=19):
        super().__init__()
        self.D = _Box()
        for idx, (o, c) in (("0", (ndf, input_nc)), ("2", (ndf // 2, ndf))):
            box = _Box()
            box.weight = nn.Parameter(torch.empty(o, c, 3, 3))
            box.bias = nn.Parameter(torch.empty(o))
            self.D.add_module(idx, box)
        for name in ("cls1", "cls2"):
            box = _Box()
            box.weight = nn.Parameter(torch.empty(num_classes, ndf // 2, 3, 3))
            box.bias = nn.Parameter(torch.empty(num_classes))
            self.add_module(name, box)
        for p in self.parameters():                       # nn.Conv2d default init
            if p.dim() == 4:
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
            else:
                nn.init.uniform_(p, -0.05, 0.05)

    def forward(self, x, size=None):
        d0, d2 = getattr(self.D, "0"), getattr(self.D, "2")
        y = F.leaky_relu(F.conv2d(x, d0.weight, d0.bias, 1, 1), 0.2)
        y = F.leaky_relu(F.conv2d(y, d2.weight, d2.bias, 1, 1), 0.2)
        out = torch.cat((F.conv2d(y, self.cls1.weight, self.cls1.bias, 1, 1), F.conv2d(y, self.cls2.weight, self.cls2.bias, 1, 1)), 1)
        if size is not None:
            out = F.interpolate(out, size=size, mode="bilinear", align_corners=True)
        return out


def ref_soft_label_cross_entropy(pred, soft_label):
    """reference core/utils/utility.py:172-177 (pixel_weights=None): mean over pixels of -sum_c soft * log_softmax(pred)."""
    return torch.mean(torch.sum(-soft_label.float() * F.log_softmax(pred, dim=1), dim=1))


def ref_fada_step(fe, cls, model_d, opt_fea, opt_cls, opt_d, src_x, src_y, tgt_x, it, max_iter, base_lr, base_lr_d, power=0.9):
    """reference core/combos/aspp_fada.py:66-127 restated (one iteration; `it` is the already incremented iteration, :68)."""
    lr = base_lr * ((1 - float(it) / max_iter) ** power)
    lr_d = base_lr_d * ((1 - float(it) / max_iter) ** power)
    for g in opt_fea.param_groups:
        g["lr"] = lr
    for g in opt_cls.param_groups:
        g["lr"] = lr * 10
    for g in opt_d.param_groups:
        g["lr"] = lr_d
    opt_fea.zero_grad()
    opt_cls.zero_grad()
    opt_d.zero_grad()
    src_y = src_y.long()
    src_size, tgt_size = src_x.shape[-2:], tgt_x.shape[-2:]
    T = 1.8
    src_fea = fe(src_x)
    src_pred = cls(src_fea, src_size).div(T)
    loss_seg = F.cross_entropy(src_pred, src_y, ignore_index=255)
    loss_seg.backward()
    src_soft = F.softmax(src_pred, dim=1).detach()
    src_soft[src_soft > 0.9] = 0.9
    tgt_fea = fe(tgt_x)
    tgt_pred = cls(tgt_fea, tgt_size).div(T)
    tgt_soft = F.softmax(tgt_pred, dim=1).detach()
    tgt_soft[tgt_soft > 0.9] = 0.9
    tgt_d = model_d(tgt_fea, tgt_size)
    loss_adv_tgt = 0.001 * ref_soft_label_cross_entropy(tgt_d, torch.cat((tgt_soft, torch.zeros_like(tgt_soft)), 1))
    loss_adv_tgt.backward()
    opt_fea.step()
    opt_cls.step()
    opt_d.zero_grad()
    src_d = model_d(src_fea.detach(), src_size)
    loss_d_src = 0.5 * ref_soft_label_cross_entropy(src_d, torch.cat((src_soft, torch.zeros_like(src_soft)), 1))
    loss_d_src.backward()
    tgt_d = model_d(tgt_fea.detach(), tgt_size)
    loss_d_tgt = 0.5 * ref_soft_label_cross_entropy(tgt_d, torch.cat((torch.zeros_like(tgt_soft), tgt_soft), 1))
    loss_d_tgt.backward()
    opt_d.step()
    return dict(loss_seg=loss_seg.item(), loss_adv_tgt=loss_adv_tgt.item(), loss_D_src=loss_d_src.item(), loss_D_tgt=loss_d_tgt.item(),
                lr=lr, lr_d=lr_d)
