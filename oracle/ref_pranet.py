"""ORACLE (test infrastructure - never imported by the product path): CPU restatement of the PraNet pieces of SURVEY 8f row N3,
pinned by fixtures generated from the reference's own code (oracle/make_golden.py: g11_*).

  structure_loss     reference core/trainers/pranet_trainer.py:22-31
  ConvBN, RFB, PartialDecoder, Bottle2neck, Res2NetTrunk, PraNet     reference core/models/classifiers/pranet/PraNet_Res2Net.py:7-179,
                                                                     Res2Net_v1b.py:16-164 (see the table further down)
"""
import numpy as np
import torch


def box31(mask):
    """F.avg_pool2d(mask, kernel_size=31, stride=1, padding=15) with torch's default count_include_pad=True: the zero padding is part of
    every window, so each output is (sum over the in-image part of the 31x31 window) / 961.  Integral image in float64."""
    m = np.asarray(mask, np.float64)
    B, C, H, W = m.shape
    ii = np.zeros((B, C, H + 1, W + 1))
    ii[:, :, 1:, 1:] = m.cumsum(2).cumsum(3)
    h0, h1 = np.clip(np.arange(H) - 15, 0, H), np.clip(np.arange(H) + 16, 0, H)
    w0, w1 = np.clip(np.arange(W) - 15, 0, W), np.clip(np.arange(W) + 16, 0, W)
    s = ii[:, :, h1][:, :, :, w1] - ii[:, :, h0][:, :, :, w1] - ii[:, :, h1][:, :, :, w0] + ii[:, :, h0][:, :, :, w0]
    return s / 961.0


def structure_loss(pred, mask):
    """pranet_trainer.py:22-31.  `F.binary_cross_entropy_with_logits(pred, mask, reduce='none')`: the string is truthy, so torch's legacy
    argument handling turns it into reduction='mean' - the BCE term is ONE scalar, the mean over every pixel of the batch, and the
    per-image weighting `(weit * wbce).sum / weit.sum` returns that scalar again.  Only the IoU term is weighted.
    pred, mask: torch tensors [B,1,H,W]; differentiable in pred (float64 arithmetic)."""
    x = pred.double()
    z = mask.double()
    weit = 1.0 + 5.0 * (torch.from_numpy(box31(mask.detach().numpy())) - z).abs()
    bce = torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-x.abs()))
    wbce = bce.mean()
    p = torch.sigmoid(x)
    inter = (p * z * weit).sum(dim=(2, 3))
    union = ((p + z) * weit).sum(dim=(2, 3))
    wiou = 1.0 - (inter + 1.0) / (union - inter + 1.0)
    return (wbce + wiou).mean()


# ------------------------------------------------------------------------------------------------ the network (oracle restatement)
#   ConvBN          reference BasicConv2d            core/models/classifiers/pranet/PraNet_Res2Net.py:7-20  (conv + BatchNorm2d, NO activation:
#                                                     the ReLU member of the reference class is never applied)
#   RFB             reference RFB_modified           PraNet_Res2Net.py:23-59
#   PartialDecoder  reference aggregation            PraNet_Res2Net.py:62-95
#   Bottle2neck     reference Bottle2neck            core/models/classifiers/pranet/Res2Net_v1b.py:16-92
#   Res2NetTrunk    reference Res2Net (v1b stem)     Res2Net_v1b.py:95-164 (the classifier head fc is kept for state_dict parity, never run)
#   PraNet          reference PraNet                 PraNet_Res2Net.py:98-179
# Attribute names equal the reference's, so state_dict keys (922) and their order are the reference's (tests/golden/g8_pranet_keys.json).
import math

import torch.nn as nn
import torch.nn.functional as F


class ConvBN(nn.Module):
    def __init__(self, cin, cout, kernel_size, padding=0, dilation=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, 1, padding, dilation, bias=False)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return self.bn(self.conv(x))


class RFB(nn.Module):
    """Four branches (1x1; then 1xk, kx1 and a 3x3 with dilation k for k = 3, 5, 7), concatenated, fused by a 3x3, plus a 1x1 shortcut."""

    def __init__(self, cin, c):
        super().__init__()
        self.branch0 = nn.Sequential(ConvBN(cin, c, 1))
        for i, k in ((1, 3), (2, 5), (3, 7)):
            setattr(self, "branch%d" % i, nn.Sequential(ConvBN(cin, c, 1), ConvBN(c, c, (1, k), (0, k // 2)), ConvBN(c, c, (k, 1), (k // 2, 0)),
                                                       ConvBN(c, c, 3, k, k)))
        self.conv_cat = ConvBN(4 * c, c, 3, 1)
        self.conv_res = ConvBN(cin, c, 1)

    def forward(self, x):
        cat = torch.cat([getattr(self, "branch%d" % i)(x) for i in range(4)], 1)
        return F.relu(self.conv_cat(cat) + self.conv_res(x))


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)          # nn.Upsample(scale_factor=2, 'bilinear', align_corners=True)


class PartialDecoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        for i in (1, 2, 3, 4):
            setattr(self, "conv_upsample%d" % i, ConvBN(c, c, 3, 1))
        self.conv_upsample5 = ConvBN(2 * c, 2 * c, 3, 1)
        self.conv_concat2 = ConvBN(2 * c, 2 * c, 3, 1)
        self.conv_concat3 = ConvBN(3 * c, 3 * c, 3, 1)
        self.conv4 = ConvBN(3 * c, 3 * c, 3, 1)
        self.conv5 = nn.Conv2d(3 * c, 1, 1)

    def forward(self, x1, x2, x3):
        """x1 coarsest (1/32), x2 (1/16), x3 (1/8)."""
        x2_1 = self.conv_upsample1(_up2(x1)) * x2
        x3_1 = self.conv_upsample2(_up2(_up2(x1))) * self.conv_upsample3(_up2(x2)) * x3
        x2_2 = self.conv_concat2(torch.cat((x2_1, self.conv_upsample4(_up2(x1))), 1))
        x3_2 = self.conv_concat3(torch.cat((x3_1, self.conv_upsample5(_up2(x2_2))), 1))
        return self.conv5(self.conv4(x3_2))


class Bottle2neck(nn.Module):
    """Res2Net bottleneck: 1x1 to `scale` groups of `width` channels; groups 0..scale-2 go through 3x3 convs, each (in a 'normal' block)
    receiving the previous group's output added to its input; the last group passes through unchanged ('normal') or through a 3x3
    average pool ('stage', the first block of a stage, where every group takes its own input and the 3x3s carry the stride)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, baseWidth=26, scale=4, stype="normal"):
        super().__init__()
        width = int(math.floor(planes * (baseWidth / 64.0)))
        self.conv1 = nn.Conv2d(inplanes, width * scale, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width * scale)
        self.nums = 1 if scale == 1 else scale - 1
        self.convs = nn.ModuleList([nn.Conv2d(width, width, 3, stride, 1, bias=False) for _ in range(self.nums)])
        self.bns = nn.ModuleList([nn.BatchNorm2d(width) for _ in range(self.nums)])
        self.conv3 = nn.Conv2d(width * scale, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.downsample = downsample
        self.stride, self.stype, self.scale, self.width = stride, stype, scale, width

    def forward(self, x):
        groups = torch.split(F.relu(self.bn1(self.conv1(x))), self.width, 1)
        outs, prev = [], None
        for i in range(self.nums):
            inp = groups[i] if (i == 0 or self.stype == "stage") else prev + groups[i]
            prev = F.relu(self.bns[i](self.convs[i](inp)))
            outs.append(prev)
        if self.scale != 1:
            last = groups[self.nums]
            outs.append(F.avg_pool2d(last, 3, self.stride, 1) if self.stype == "stage" else last)
        y = self.bn3(self.conv3(torch.cat(outs, 1)))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class Res2NetTrunk(nn.Module):
    def __init__(self, layers=(3, 4, 6, 3), baseWidth=26, scale=4, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(3, 32, 3, 2, 1, bias=False), nn.BatchNorm2d(32), nn.ReLU(), nn.Conv2d(32, 32, 3, 1, 1, bias=False),
                                   nn.BatchNorm2d(32), nn.ReLU(), nn.Conv2d(32, 64, 3, 1, 1, bias=False))
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for li, (planes, n, stride) in enumerate(zip((64, 128, 256, 512), layers, (1, 2, 2, 2)), 1):
            blocks = []
            for b in range(n):
                ds = None
                if b == 0 and (stride != 1 or inplanes != planes * 4):
                    ds = nn.Sequential(nn.AvgPool2d(stride, stride, ceil_mode=True, count_include_pad=False), nn.Conv2d(inplanes, planes * 4, 1, bias=False),
                                       nn.BatchNorm2d(planes * 4))
                blocks.append(Bottle2neck(inplanes, planes, stride if b == 0 else 1, ds, baseWidth, scale, "stage" if b == 0 else "normal"))
                inplanes = planes * 4
            setattr(self, "layer%d" % li, nn.Sequential(*blocks))
        self.fc = nn.Linear(512 * 4, num_classes)          # present in the reference's state_dict; PraNet never calls it

    def stem(self, x):
        return F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)


class PraNet(nn.Module):
    """Res2Net-50 features at 1/8, 1/16, 1/32 -> RFB (32 channels each) -> partial decoder -> global map; three reverse-attention
    branches refine it from coarse to fine.  Returns the four side outputs upsampled to the input size, coarse decoder map first
    (lateral_map_5, 4, 3, 2 of PraNet_Res2Net.py:179)."""

    def __init__(self, channel=32):
        super().__init__()
        self.resnet = Res2NetTrunk()
        self.rfb2_1, self.rfb3_1, self.rfb4_1 = RFB(512, channel), RFB(1024, channel), RFB(2048, channel)
        self.agg1 = PartialDecoder(channel)
        self.ra4_conv1 = ConvBN(2048, 256, 1)
        for i in (2, 3, 4):
            setattr(self, "ra4_conv%d" % i, ConvBN(256, 256, 5, 2))
        self.ra4_conv5 = ConvBN(256, 1, 1)
        for lvl, cin in ((3, 1024), (2, 512)):
            setattr(self, "ra%d_conv1" % lvl, ConvBN(cin, 64, 1))
            setattr(self, "ra%d_conv2" % lvl, ConvBN(64, 64, 3, 1))
            setattr(self, "ra%d_conv3" % lvl, ConvBN(64, 64, 3, 1))
            setattr(self, "ra%d_conv4" % lvl, ConvBN(64, 1, 3, 1))

    @staticmethod
    def _resize(x, factor):
        return F.interpolate(x, scale_factor=factor, mode="bilinear")          # align_corners left at its default (False), as in the reference

    def _tap(self, name, t):
        """Test infrastructure: with a `_taps` dict on the module, named intermediates are kept (and their gradients retained) - the teacher-forced
        tests feed them to the engine block by block."""
        taps = self.__dict__.get("_taps")
        if taps is not None:
            if t.requires_grad:
                t.retain_grad()
            taps[name] = t
        return t

    def forward(self, x):
        r, tap = self.resnet, self._tap
        y = tap("stem0", r.conv1[2](r.conv1[1](r.conv1[0](x))))
        y = tap("stem1", r.conv1[5](r.conv1[4](r.conv1[3](y))))
        y = tap("stem", F.max_pool2d(F.relu(r.bn1(r.conv1[6](y))), 3, 2, 1))          # = r.stem(x)
        ends = {}
        for li in (1, 2, 3, 4):
            for bi, blk in enumerate(getattr(r, "layer%d" % li)):
                y = tap("resnet.layer%d.%d" % (li, bi), blk(y))
            ends[li] = y
        x2, x3, x4 = ends[2], ends[3], ends[4]
        f2, f3, f4 = tap("rfb2", self.rfb2_1(x2)), tap("rfb3", self.rfb3_1(x3)), tap("rfb4", self.rfb4_1(x4))
        coarse = tap("coarse", self.agg1(f4, f3, f2))                                  # 1/8 resolution, one channel
        maps = [self._resize(coarse, 8)]
        # reverse attention, level 4 (1/32): erase what the coarser map already marks as foreground, predict a residual
        g = self._resize(coarse, 0.25)
        y = self.ra4_conv1((1 - torch.sigmoid(g)) * x4)
        y = F.relu(self.ra4_conv4(F.relu(self.ra4_conv3(F.relu(self.ra4_conv2(y))))))
        g = tap("ra4", self.ra4_conv5(y) + g)
        maps.append(self._resize(g, 32))
        for lvl, feat, up in ((3, x3, 16), (2, x2, 8)):
            g = self._resize(g, 2)
            y = getattr(self, "ra%d_conv1" % lvl)((1 - torch.sigmoid(g)) * feat)
            y = F.relu(getattr(self, "ra%d_conv3" % lvl)(F.relu(getattr(self, "ra%d_conv2" % lvl)(y))))
            g = tap("ra%d" % lvl, getattr(self, "ra%d_conv4" % lvl)(y) + g)
            maps.append(self._resize(g, up))
        return tuple(tap("map%d" % i, m) for i, m in enumerate(maps))
