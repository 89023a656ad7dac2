"""ORACLE (test infrastructure - never imported by the product path): CPU restatement of the GALD / GCPA path of SURVEY 8f row N4,
pinned by fixtures generated from the reference's own modules (oracle/make_golden.py: g13_gald_modules, g13_gald_224, g8_gald_keys.json).

  ConvBN6, HarDBlock, HarDNet68     reference core/models/classifiers/gcpacc/encoders/hardnet_68.py:56-80, 83-160, 163-262
  CrissCross                        reference core/models/classifiers/gcpacc/contextagg/ccnet.py:37-127
  LocalAtten                        reference core/models/classifiers/gcpacc/contextagg/GALDNet.py:124-157
  FAM                               reference core/models/classifiers/gcpacc/gcpa_gald.py:47-107
  GCPAEncoder, GCPADecoder          reference core/models/classifiers/gcpacc/gcpa_cc2.py:16-83
  gald_losses                       reference core/trainers/gald_trainer.py:66-84

Attribute names equal the reference's, so the state_dict keys (404 + 186) and their order are the reference's.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _tap(module, name, t):
    """Test infrastructure: with a `_taps` dict on the module, named intermediates are kept (and their gradients retained) - the teacher-forced tests
    feed them to the engine block by block."""
    taps = module.__dict__.get("_taps")
    if taps is not None:
        if t.requires_grad:
            t.retain_grad()
        taps[name] = t
    return t


class ConvBN6(nn.Module):
    """conv (no bias, pad k // 2) -> BatchNorm2d -> ReLU6"""

    def __init__(self, cin, cout, kernel=3, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel, stride, kernel // 2, bias=False)
        self.norm = nn.BatchNorm2d(cout)

    def forward(self, x):
        return F.relu6(self.norm(self.conv(x)))


def hard_link(layer, base_ch, growth, mul):
    """Harmonic dense connectivity: layer n takes the outputs of layers n - 2^i for every 2^i dividing n (layer 0 is the block input);
    its width is growth * mul^(number of links beyond the first), rounded to an even number.  Returns (out_ch, in_ch, links)."""
    if layer == 0:
        return base_ch, 0, []
    links, width = [], float(growth)
    for i in range(10):
        step = 1 << i
        if layer % step == 0:
            links.append(layer - step)
            if i:
                width *= mul
    out_ch = int(int(width + 1) / 2) * 2
    return out_ch, sum(hard_link(j, base_ch, growth, mul)[0] for j in links), links


class HarDBlock(nn.Module):
    def __init__(self, cin, growth, mul, n_layers):
        super().__init__()
        self.links, convs, self.out_channels = [], [], 0
        for i in range(n_layers):
            out_ch, in_ch, links = hard_link(i + 1, cin, growth, mul)
            self.links.append(links)
            convs.append(ConvBN6(in_ch, out_ch))
            if i % 2 == 0 or i == n_layers - 1:
                self.out_channels += out_ch
        self.layers = nn.ModuleList(convs)

    def forward(self, x):
        outs = [x]
        for conv, links in zip(self.layers, self.links):
            inp = torch.cat([outs[j] for j in links], 1) if len(links) > 1 else outs[links[0]]
            outs.append(conv(inp))
        n = len(outs)
        return torch.cat([outs[i] for i in range(n) if i == n - 1 or i % 2 == 1], 1)          # the last layer and the odd ones


class HarDNet68(nn.Module):
    """base: conv 3->32 /2, conv 32->64, max-pool 3/2/1, then five (HarDBlock, 1x1 transition[, max-pool 2/2]) groups; the four
    outputs the decoder uses are the transitions at 1/4 (128), 1/8 (320), 1/16 (640) and 1/32 (1024).  The ImageNet head is kept for
    state_dict parity and never run."""

    def __init__(self):
        super().__init__()
        mods = [ConvBN6(3, 32, 3, 2), ConvBN6(32, 64, 3), nn.MaxPool2d(3, 2, 1)]
        ch = 64
        self.taps = []
        for width, growth, n, down in zip((128, 256, 320, 640, 1024), (14, 16, 20, 40, 160), (8, 16, 16, 16, 4), (1, 0, 1, 1, 0)):
            blk = HarDBlock(ch, growth, 1.7, n)
            mods += [blk, ConvBN6(blk.out_channels, width, 1)]
            if width != 256:
                self.taps.append(len(mods) - 1)
            ch = width
            if down:
                mods.append(nn.MaxPool2d(2, 2))
        mods.append(nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), nn.Dropout(0.1), nn.Linear(ch, 1000)))
        self.base = nn.ModuleList(mods)

    def forward(self, x):
        outs = []
        for i, m in enumerate(self.base[:-1]):
            x = _tap(self, "base.%d" % i, m(x))
            if i in self.taps:
                outs.append(x)
        return outs


class CrissCross(nn.Module):
    """Every pixel attends to its column and its row: affinities q . k along both, the column's own position masked with -inf (the row already
    holds it), ONE softmax over the H + W candidates, values aggregated with those weights; gamma * aggregate + x."""

    def __init__(self, c):
        super().__init__()
        self.query_conv = nn.Conv2d(c, c // 8, 1)
        self.key_conv = nn.Conv2d(c, c // 8, 1)
        self.value_conv = nn.Conv2d(c, c, 1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        q, k, v = self.query_conv(x), self.key_conv(x), self.value_conv(x)
        H = x.shape[2]
        e_col = torch.einsum("bchw,bcgw->bhwg", q, k)                                       # [b, h, w, h']
        e_col = e_col.masked_fill(torch.eye(H, dtype=torch.bool, device=x.device).view(1, H, 1, H), float("-inf"))
        e_row = torch.einsum("bchw,bchv->bhwv", q, k)                                       # [b, h, w, w']
        att = torch.softmax(torch.cat([e_col, e_row], 3), 3)
        out = torch.einsum("bcgw,bhwg->bchw", v, att[..., :H]) + torch.einsum("bchv,bhwv->bchw", v, att[..., H:])
        return self.gamma * out + x


class LocalAtten(nn.Module):
    """x + x * sigmoid(up(dw2(dw1(x)))): two depthwise 3x3 stride-2 convs WITHOUT padding (bias, BatchNorm, ReLU each), bilinear
    align_corners=True back to the input size.  dconv3 exists in the reference's state_dict and is never run."""

    def __init__(self, c):
        super().__init__()
        for i in (1, 2, 3):
            setattr(self, "dconv%d" % i, nn.Sequential(nn.Conv2d(c, c, 3, 2, groups=c), nn.BatchNorm2d(c), nn.ReLU()))

    def forward(self, x):
        g = F.interpolate(self.dconv2(self.dconv1(x)), size=x.shape[2:], mode="bilinear", align_corners=True)
        return x + x * torch.sigmoid(g)


class FAM(nn.Module):
    """Feature aggregation of an encoder feature (left), the coarser decoder feature (down) and the context feature (right): three
    conv(bias)-BN-ReLU stems, three gated products at left's resolution, a fusing conv-BN-ReLU."""

    def __init__(self, c_left, c_down, c_right, c=256):
        super().__init__()
        self.conv0, self.bn0 = nn.Conv2d(c_left, c, 3, 1, 1), nn.BatchNorm2d(c)
        self.conv1, self.bn1 = nn.Conv2d(c_down, c, 3, 1, 1), nn.BatchNorm2d(c)
        self.conv2, self.bn2 = nn.Conv2d(c_right, c, 3, 1, 1), nn.BatchNorm2d(c)
        self.conv_d1 = nn.Conv2d(c, c, 3, 1, 1)
        self.conv_d2 = nn.Conv2d(c, c, 3, 1, 1)
        self.conv_l = nn.Conv2d(c, c, 3, 1, 1)
        self.conv3, self.bn3 = nn.Conv2d(3 * c, c, 3, 1, 1), nn.BatchNorm2d(c)

    def forward(self, left, down, right):
        left = F.relu(self.bn0(self.conv0(left)))
        down = F.relu(self.bn1(self.conv1(down)))
        right = F.relu(self.bn2(self.conv2(right)))
        size = left.shape[2:]
        fit = lambda t: t if t.shape[2:] == size else F.interpolate(t, size=size, mode="bilinear")          # align_corners False
        z1 = F.relu(self.conv_l(left) * fit(down))
        z2 = F.relu(fit(self.conv_d1(down)) * left)
        z3 = F.relu(fit(self.conv_d2(right)) * left)
        return F.relu(self.bn3(self.conv3(torch.cat((z1, z2, z3), 1))))


class GCPAEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.hardnet = HarDNet68()

    def forward(self, x):
        self.hardnet.__dict__["_taps"] = self.__dict__.get("_taps")
        return self.hardnet(x)


class GCPADecoder(nn.Module):
    def __init__(self, num_classes=19, c=256):
        super().__init__()
        self.fam45, self.fam34, self.fam23 = FAM(640, c, c, c), FAM(320, c, c, c), FAM(128, c, c, c)
        for i in (5, 4, 3, 2):
            setattr(self, "linear%d" % i, nn.Conv2d(c, num_classes, 3, 1, 1))
        self.conva = nn.Sequential(nn.Conv2d(1024, c, 3, padding=1, bias=False), nn.BatchNorm2d(c), nn.ReLU())
        self.long_relation = CrissCross(c)
        self.local_attention_4, self.local_attention_3, self.local_attention_2 = LocalAtten(c), LocalAtten(c), LocalAtten(c)

    def forward(self, x, feats):
        f2, f3, f4, f5 = feats
        tap = lambda name, t: _tap(self, name, t)
        top = tap("conva", self.conva(f5))
        ctx = tap("cca2", self.long_relation(tap("cca1", self.long_relation(top))))      # the SAME criss-cross module twice (recurrence 2)
        o4 = tap("fam45", self.fam45(f4, top, tap("lam4", self.local_attention_4(ctx))))
        o3 = tap("fam34", self.fam34(f3, o4, tap("lam3", self.local_attention_3(ctx))))
        o2 = tap("fam23", self.fam23(f2, o3, tap("lam2", self.local_attention_2(ctx))))
        up = lambda t: F.interpolate(t, size=x.shape[2:], mode="bilinear")
        lows = [tap("linear%d" % i, getattr(self, "linear%d" % i)(v)) for i, v in ((5, top), (4, o4), (3, o3), (2, o2))]
        return tuple(tap("out%d" % i, up(v)) for i, v in enumerate(lows))


def gald_losses(outs, label, ignore_index=255):
    """The four deep-supervision cross-entropies (out5, out4, out3, out2) and their weighted sum 0.4 / 0.6 / 0.8 / 1."""
    ls = [F.cross_entropy(o, label, ignore_index=ignore_index) for o in outs]
    return ls, ls[3] * 1 + ls[2] * 0.8 + ls[1] * 0.6 + ls[0] * 0.4
