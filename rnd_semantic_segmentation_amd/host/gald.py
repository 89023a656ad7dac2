"""The GALD / GCPA path on the MI355X engine: HarDNet-68 encoder, criss-cross attention, local attention modules, feature
aggregation modules, four deep-supervision heads - forward AND backward as a tape of C-ABI launches (csrc/gconv.hip, gnet.hip, gald.hip).

  GCPAEncoder, GCPADecoder     reference core/models/classifiers/gcpacc/gcpa_cc2.py:16-83 (state_dict keys: 404 + 186)
  HarDNet-68                   reference core/models/classifiers/gcpacc/encoders/hardnet_68.py:56-262
  CrissCrossAttention          reference core/models/classifiers/gcpacc/contextagg/ccnet.py:37-127
  LocalAttenModule             reference core/models/classifiers/gcpacc/contextagg/GALDNet.py:124-157
  FAM                          reference core/models/classifiers/gcpacc/gcpa_gald.py:47-107
  GALDTrainer                  reference core/trainers/gald_trainer.py:13-112

Same engine as host/pranet.py (NHWC bf16 activations, concatenations as channel-slice views where the link pattern allows, BatchNorm2d on
batch statistics from the conv's tile sums, one flat fp32 parameter / gradient buffer per module, Adam in one launch).  What is new here:
ReLU6, max pools with a stored argmax, the depthwise stride-2 convs, the attention core (one workgroup per pixel instead of the reference's
permute / contiguous / bmm chain), parameters shared between two applications of a module (the criss-cross module runs twice: gradients
accumulate in its slots), class-logit heads upsampled with align_corners=False and a cross-entropy kernel on NHWC logits.
"""
import os
from datetime import datetime

import torch
import torch.nn as nn

from .. import _lib
from .. import gk
from .. import kernels as K
from . import arch
from .metrics import AverageMeter, adjust_learning_rate, confusion_matrix, dump_json, intersectionAndUnionGPU
from .plugin import BaseTrainer
from .pranet import FlatAdam, _acc, _Engine, _grad_target, _Run, _rup32, _tile_route, _Unit


class _GaldRun(_Run):
    WGRAD_STREAM = True           # GALD trains eagerly and its weight gradients are few and large: +1.5 % with them beside the data-gradient chain
    def maxpool(self, x, k, stride, pad):
        H, W = x.t.shape[1], x.t.shape[2]
        if self.f32:
            return self.var(gk.gpool_f32(x.t, k, stride, pad, 2), False)
        o, idx = gk.gmaxpool(x.t, k, stride, pad)
        ov = self.var(o)

        def back():
            if ov.g is not None:
                _acc(x, gk.gmaxpool_bwd(ov.g, idx, (H, W), k, stride, pad, dx=_grad_target(x)), True)
                ov.g = None
        self.record(back)
        return ov

    def dw_bn_relu(self, x, u):
        """Conv2d(C, C, 3, groups=C, stride=2) with bias -> BatchNorm2d -> ReLU (GALDNet.py:127-136); u.weight is [C,1,3,3]."""
        net, bn = self.net, u.bn
        C = u.cout
        if not self.train:
            sc, sh = net._eval_fold(u)
            if self.f32:
                return self.var(gk.gdwconv_f32(x.t, u.weight.detach(), u.bias.detach(), 2, 0, scale=sc, shift=sh, relu=True), False)
            y, _ = gk.gdwconv(x.t, u.weight.detach(), u.bias.detach(), 2, 0)
            return self.var(gk.gbn_apply(y, sc, sh, 1), False)
        y, st = gk.gdwconv(x.t, u.weight.detach(), u.bias.detach(), 2, 0, stats=True)
        M = y.shape[0] * y.shape[1] * y.shape[2]
        fin = gk.gbn_finalize(st, C, M, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
        o = gk.gbn_apply(y, fin[2], fin[3], 1)
        ov = self.var(o)

        def back():
            g = ov.g
            ov.g = None
            if g is None:
                return
            dbeta, dgamma = net._grad_of(bn.bias), net._grad_of(bn.weight)
            gk.gbn_bwd_sums(g, y, o, fin[0], fin[1], dbeta, dgamma)
            dy = gk.gbn_bwd_apply(g, y, o, fin[0], fin[1], bn.weight, dbeta, dgamma, M)
            dx = gk.gdwconv_backward(dy, x.t, u.weight.detach(), net._grad_of(u.weight), net._grad_of(u.bias), 2, 0, need_dx=x.needs)
            if dx is not None:
                _acc(x, dx, True)
        self.record(back)
        return ov

    def gate(self, x, g):
        """x + x * sigmoid(g) (GALDNet.py:150-157)"""
        if self.f32:
            return self.var(gk.gpoint_f32(gk.PW_GATE, x.t, g.t), False)
        ov = self.var(gk.ggate(x.t, g.t))

        def back():
            if ov.g is None:
                return
            dx, dg = gk.ggate_bwd(x.t, g.t, ov.g)
            ov.g = None
            _acc(x, dx, True)
            _acc(g, dg, True)
        self.record(back)
        return ov

    def mulrelu(self, a, b, out=None):
        """relu(a * b) (gcpa_gald.py:88-101)"""
        if self.f32:
            return self.var(gk.gpoint_f32(gk.PW_MULRELU, a.t, b.t, out=out), False)
        o = gk.gbinary(gk.OP_MULRELU, a.t, b.t, out=out)
        ov = self.var(o)

        def back():
            g = ov.g
            ov.g = None
            if g is None:
                return
            gm = gk.gbinary(gk.OP_RELU_MASK, g, o)
            _acc(a, gk.gbinary(gk.OP_MUL, gm, b.t), True)
            _acc(b, gk.gbinary(gk.OP_MUL, gm, a.t), True)
        self.record(back)
        return ov

    def ce_head(self, low, labels, ignore_index):
        """criterion(F.interpolate(low, size=labels.shape[-2:], mode="bilinear"), labels) (gcpa_cc2.py:78-81 + gald_trainer.py:76-79) fused: the
        full-resolution logits are never written; d loss / d low comes out of the same pass."""
        loss_out, dlow = K.upsample_ce(low.t, labels, want_grad=self.rec, ignore_index=ignore_index, align_corners=False)
        _count_bad_labels(self.net, loss_out)
        ov = self.var(loss_out[0].clone())

        def back():
            if ov.g is not None:
                _acc(low, dlow * ov.g, True)
                ov.g = None
        self.record(back)
        return ov

    def criss_cross(self, x, uq, uk, uv, gamma):
        """CrissCrossAttention.forward (ccnet.py:56-127): gamma * aggregate + x."""
        net = self.net
        q, k, v = self.conv_bias(x, uq, out_f32=False), self.conv_bias(x, uk, out_f32=False), self.conv_bias(x, uv, out_f32=False)
        if self.f32:
            C = v.t.shape[-1]
            return self.var(gk.gpoint_f32(gk.PW_AFFINE, gk.gcca_f32(q.t, k.t, v.t), x.t, scale=gamma.detach().expand(C).contiguous(), shift=net._zeros(C)), False)
        agg, att = gk.gcca_fwd(q.t, k.t, v.t)
        C = agg.shape[-1]
        gvec = gamma.detach().expand(C).contiguous()
        zero = net._zeros(C)
        ov = self.var(gk.gbn_apply(agg, gvec, zero, 0, add=x.t))

        def back():
            g = ov.g
            ov.g = None
            if g is None:
                return
            _acc(x, g, False)                                              # the residual
            dagg = gk.gbn_apply(g, gvec, zero, 0)
            per_c = torch.empty(C, dtype=torch.float32, device=g.device)
            gk.gbn_bwd_sums(g, agg, None, zero, net._ones(C), torch.empty_like(per_c), per_c)      # sum_m g[m][c] * agg[m][c]
            slot, acc = net._grad_slot(gamma)
            s = per_c.sum().reshape(1)
            slot.add_(s) if acc else slot.copy_(s)
            dq, dk, dv = gk.gcca_bwd(q.t, k.t, v.t, att, dagg)
            _acc(q, dq, True)
            _acc(k, dk, True)
            _acc(v, dv, True)
        self.record(back)
        return ov


def _count_bad_labels(holder, loss_out):
    """Labels outside [0, K) that are not ignore_index are skipped AND counted by the cross-entropy kernels (loss_out[2]); torch's
    CrossEntropyLoss (gald_trainer.py:107) would device-assert on them.  Every call's count goes into a persistent device counter on `holder`
    (the decoder / the criterion module), read by GALDTrainer where it fetches the loss anyway."""
    bad = holder.__dict__.get("bad_labels")
    if bad is None or bad.device != loss_out.device:
        bad = holder.__dict__["bad_labels"] = torch.zeros(1, dtype=torch.float32, device=loss_out.device)
    bad.add_(loss_out[2:3])


def take_bad_labels(*holders):
    """Sum and reset the counters of `_count_bad_labels` (one host sync); under torch.distributed the sum is all-reduced first so that every rank
    sees the same count and raises (or not) together."""
    bads = [h.__dict__.get("bad_labels") for h in holders]
    bads = [b for b in bads if b is not None]
    if not bads:
        return 0
    total = torch.stack([b.reshape(()) for b in bads]).sum().reshape(1)
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        torch.distributed.all_reduce(total)
    n = int(total.item())
    for b in bads:
        b.zero_()
    return n


# ------------------------------------------------------------------------------------------------ HarDNet-68
def _hard_link(layer, base_ch, growth, mul):
    """Harmonic dense links (hardnet_68.py:84-102): layer n reads layers n - 2^i for every 2^i that divides n; width growth * mul^(links - 1),
    rounded to even.  -> (out channels, in channels, links)"""
    if layer == 0:
        return base_ch, 0, []
    links, width = [], float(growth)
    for i in range(10):
        if layer % (1 << i) == 0:
            links.append(layer - (1 << i))
            if i:
                width *= mul
    return int(int(width + 1) / 2) * 2, sum(_hard_link(j, base_ch, growth, mul)[0] for j in links), links


def _conv_layer(name, cin, cout, k=3, stride=1):
    return _Unit(name + ".conv", name + ".norm", cin, cout, k, stride, k // 2)


def _hard_block(run, x, layers, links, out_ch):
    """HarDBlock.forward (hardnet_68.py:137-160): the layers that make up the block's output (the odd ones and the last) write straight
    into their channel range of the output buffer; a layer with several links reads a gathered copy."""
    B, H, W, _ = x.t.shape
    n = len(layers)
    keep = [i for i in range(1, n + 1) if i == n or i % 2 == 1]                 # indices into outs (0 = the block input)
    offs, off = {}, 0
    for i in keep:
        offs[i] = off
        off += layers[i - 1].cout
    outbuf = gk.new(B, H, W, out_ch, x.t.device, x.t.dtype)
    # the gather buffers of the layers with several links, allocated up front: a layer's BatchNorm apply writes its output into the block's buffer AND into
    # the channel range it has in every later layer's gather (up to gk.APPLY_MAX_EXTRA of them; round 5) - only the block input is still copied
    gathers = {}
    for li, (u, lk) in enumerate(zip(layers, links), 1):
        if len(lk) > 1:
            widths = [x.t.shape[-1] if j == 0 else layers[j - 1].cout for j in lk]
            cin = sum(widths)
            # a layer the MFMA-tile kernels take (pranet._tile_route: the big gathered layers, 466 -> 168 ...) reads a gather buffer of the 32-padded
            # channel count, pad channels zero: the kernels' operand as it is
            cbuf = _rup32(cin) if (not run.f32 and _tile_route(u, B * H * W)) else cin
            buf = gk.new(B, H, W, cbuf, x.t.device, x.t.dtype)
            if cbuf != cin:
                buf[..., cin:].zero_()
            offs_j, o = {}, 0
            for j, c in zip(lk, widths):
                offs_j[j] = (o, c)
                o += c
            gathers[li] = (buf, offs_j)
    outs, fused = [x], set()
    for li, (u, lk) in enumerate(zip(layers, links), 1):
        if len(lk) > 1:
            buf, offs_j = gathers[li]
            pieces = []
            for j in lk:
                o, c = offs_j[j]
                pieces.append(run.alias_into(outs[j], buf[..., o:o + c]) if (j, li) in fused else run.copy_into(outs[j], buf[..., o:o + c]))
            inp = run.cat(buf, pieces)
        else:
            inp = outs[lk[0]]
        dst = outbuf[..., offs[li]:offs[li] + u.cout] if li in offs else None
        extras = []
        if run.multi:
            for l2 in sorted(gathers):
                if l2 > li and li in gathers[l2][1] and len(extras) < gk.APPLY_MAX_EXTRA:
                    o, c = gathers[l2][1][li]
                    extras.append((0, u.cout, gathers[l2][0][..., o:o + c], None))
                    fused.add((li, l2))
        outs.append(run.conv_bn(inp, u, 6, out=dst, extras=extras or None))
    return run.cat(outbuf, [outs[i] for i in keep])


def _hard_block_units(prefix, cin, growth, mul, n):
    """(units, links, out channels) of HarDBlock(cin, growth, mul, n) with state_dict keys <prefix>layers.<i>.{conv,norm}.*"""
    layers, links, out_ch = [], [], 0
    for i in range(n):
        oc, ic, lk = _hard_link(i + 1, cin, growth, mul)
        layers.append(_conv_layer("%slayers.%d" % (prefix, i), ic, oc))
        links.append(lk)
        if i % 2 == 0 or i == n - 1:
            out_ch += oc
    return layers, links, out_ch


class GCPAEncoder(_Engine):
    """HarDNet-68 trunk (gcpa_cc2.py:16-23); forward(x [B,3,H,W]) -> the four features at 1/4 (128), 1/8 (320), 1/16 (640), 1/32 (1024),
    bf16, NCHW-shaped views of NHWC memory.  The reference loads 'pretrained/hardnet68.pth', which the image lacks: weights keep the module
    defaults unless a checkpoint is loaded."""
    RUN = _GaldRun
    PAD_IMAGE = True

    def __init__(self):
        super().__init__()
        p = "hardnet.base."
        order = [_conv_layer(p + "0", 3, 32, 3, 2), _conv_layer(p + "1", 32, 64, 3)]
        self._seq = [("conv", order[0]), ("conv", order[1]), ("pool3", None)]
        ch, idx = 64, 3
        for width, growth, n, down in zip((128, 256, 320, 640, 1024), (14, 16, 20, 40, 160), (8, 16, 16, 16, 4), (1, 0, 1, 1, 0)):
            layers, links, out_ch = _hard_block_units("%s%d." % (p, idx), ch, growth, 1.7, n)
            order += layers
            self._seq.append(("block", (layers, links, out_ch)))
            trans = _conv_layer(p + str(idx + 1), out_ch, width, 1)
            order.append(trans)
            self._seq.append(("conv_tap" if width != 256 else "conv", trans))
            idx += 2
            ch = width
            if down:
                self._seq.append(("pool2", None))
                idx += 1
        self._head_key = p + str(idx) + ".3"
        order += [(self._head_key + ".weight", torch.empty(1000, 1024).uniform_(-1, 1) / 32.0),          # the ImageNet head: in the reference's
                  (self._head_key + ".bias", torch.empty(1000).uniform_(-1, 1) / 32.0)]                    # state_dict, never run
        self._register(order)

    def _graph(self, run, x):
        feats, y = [], x
        for i, (kind, arg) in enumerate(self._seq):          # i = the module's index in the reference's `hardnet.base` list
            if kind in ("conv", "conv_tap"):
                y = run.conv_bn(y, arg, 6)
            elif kind == "pool3":
                y = run.maxpool(y, 3, 2, 1)
            elif kind == "pool2":
                y = run.maxpool(y, 2, 2, 0)
            else:
                y = _hard_block(run, y, *arg)
            y = run.tap("base.%d" % i, y)
            if kind == "conv_tap":
                feats.append(y)
        return feats


def _fam_units(prefix, c_left, c_down, c_right, c):
    n = prefix
    u = dict(conv0=_Unit(n + "conv0", n + "bn0", c_left, c, 3, 1, 1), conv1=_Unit(n + "conv1", n + "bn1", c_down, c, 3, 1, 1),
             conv2=_Unit(n + "conv2", n + "bn2", c_right, c, 3, 1, 1), conv_d1=_Unit(n + "conv_d1", None, c, c, 3, 1, 1),
             conv_d2=_Unit(n + "conv_d2", None, c, c, 3, 1, 1), conv_l=_Unit(n + "conv_l", None, c, c, 3, 1, 1),
             conv3=_Unit(n + "conv3", n + "bn3", 3 * c, c, 3, 1, 1))
    for k in ("conv0", "conv1", "conv2", "conv3"):
        u[k].bias = True                                     # nn.Conv2d default: these convs carry a bias in front of their BatchNorm
    return u, [u[k] for k in ("conv0", "conv1", "conv2", "conv_d1", "conv_d2", "conv_l", "conv3")]


def _fam_block(run, U, c, left, down, right):
    """FAM.forward (gcpa_gald.py:83-107)."""
    left = run.conv_bn(left, U["conv0"], True)
    down = run.conv_bn(down, U["conv1"], True)
    right = run.conv_bn(right, U["conv2"], True)
    size = (left.t.shape[1], left.t.shape[2])
    fit = lambda v: v if (v.t.shape[1], v.t.shape[2]) == size else run.resize(v, None, False, size=size)
    B = left.t.shape[0]
    cat = gk.new(B, size[0], size[1], 3 * c, left.t.device, left.t.dtype)
    z1 = run.mulrelu(run.conv_bias(left, U["conv_l"], out_f32=False), fit(down), out=cat[..., :c])
    z2 = run.mulrelu(fit(run.conv_bias(down, U["conv_d1"], out_f32=False)), left, out=cat[..., c:2 * c])
    z3 = run.mulrelu(fit(run.conv_bias(right, U["conv_d2"], out_f32=False)), left, out=cat[..., 2 * c:])
    return run.conv_bn(run.cat(cat, [z1, z2, z3]), U["conv3"], True)


def _local_attention(run, x, units):
    """LocalAttenModule.forward (GALDNet.py:143-157); dconv3 is never run."""
    g = run.dw_bn_relu(run.dw_bn_relu(x, units[0]), units[1])
    return run.gate(x, run.resize(g, None, True, size=(x.t.shape[1], x.t.shape[2])))


def _lam_units(prefix, c):
    units = [_Unit("%sdconv%d.0" % (prefix, j), "%sdconv%d.1" % (prefix, j), c, c, 3, 2, 0) for j in (1, 2, 3)]
    for u in units:
        u.bias, u.depthwise = True, True
    return units


def _cca_units(prefix, c):
    """[gamma entry, query, key, value] in the reference's state_dict order (a module's own parameters precede its children's)."""
    return [(prefix + "gamma", torch.zeros(1)), _Unit(prefix + "query_conv", None, c, c // 8, 1), _Unit(prefix + "key_conv", None, c, c // 8, 1),
            _Unit(prefix + "value_conv", None, c, c, 1)]


class GCPADecoder(_Engine):
    """gcpa_cc2.py:25-83.  forward(x, feats) -> (out5, out4, out3, out2): class logits [B,19,H,W] fp32 (NCHW-shaped views of NHWC memory)."""
    RUN = _GaldRun

    def __init__(self, num_classes=19, c=256):
        super().__init__()
        self.num_classes, self._c = num_classes, c
        order, self._fam = [], {}
        for name, cl in (("fam45", 640), ("fam34", 320), ("fam23", 128)):
            self._fam[name], flat = _fam_units(name + ".", cl, c, c, c)
            order += flat
        self._lin = {i: _Unit("linear%d" % i, None, c, num_classes, 3, 1, 1) for i in (5, 4, 3, 2)}
        order += [self._lin[i] for i in (5, 4, 3, 2)]
        self._conva = _Unit("conva.0", "conva.1", 1024, c, 3, 1, 1)
        order.append(self._conva)
        cca = _cca_units("long_relation.", c)
        self._cca = cca[1:]
        order += cca
        self._lam = {}
        for i in (4, 3, 2):
            self._lam[i] = _lam_units("local_attention_%d." % i, c)
            order += self._lam[i]
        self._register(order)

    def _graph(self, run, x, f2, f3, f4, f5):
        top = run.tap("conva", run.conv_bn(f5, self._conva, True))
        gamma = arch.node_at(self, "long_relation").gamma
        ctx = run.tap("cca1", run.criss_cross(top, *self._cca, gamma))
        ctx = run.tap("cca2", run.criss_cross(ctx, *self._cca, gamma))                          # the same module twice (gcpa_cc2.py:56-57)
        c = self._c
        o4 = run.tap("fam45", _fam_block(run, self._fam["fam45"], c, f4, top, run.tap("lam4", _local_attention(run, ctx, self._lam[4]))))
        o3 = run.tap("fam34", _fam_block(run, self._fam["fam34"], c, f3, o4, run.tap("lam3", _local_attention(run, ctx, self._lam[3]))))
        o2 = run.tap("fam23", _fam_block(run, self._fam["fam23"], c, f2, o3, run.tap("lam2", _local_attention(run, ctx, self._lam[2]))))
        lows = [run.tap("linear%d" % i, run.conv_bias(v, self._lin[i])) for i, v in ((5, top), (4, o4), (3, o3), (2, o2))]
        labels = self.__dict__.get("_ce_labels")
        if labels is not None:                                                                 # the trainer's fused path: four scalar losses
            return [run.ce_head(v, labels, self._ce_ignore) for v in lows]
        size = (x.t.shape[1], x.t.shape[2])
        return [run.tap("out%d" % i, run.resize(v, None, False, size=size)) for i, v in enumerate(lows)]      # F.interpolate(..., size=x.size()[2:], mode="bilinear")

    def forward(self, x, feats):
        return super().forward(x, *feats)

    def losses(self, x, feats, labels, ignore_index=255):
        """(loss5, loss4, loss3, loss2) = criterion(out_i, labels) of gald_trainer.py:76-79 without materialising the four [B,19,H,W] logit tensors
        (upsample + cross-entropy fused, mi_upsample_ce_ex): what GALDTrainer.train_step calls."""
        self._ce_labels, self._ce_ignore = labels.long().contiguous(), int(ignore_index)
        try:
            return super().forward(x, *feats)
        finally:
            self._ce_labels = None


# ------------------------------------------------------------------------------------------------ the building blocks as stand-alone modules
# Same constructor arguments and state_dict keys as the reference's classes, the SAME graph functions the two networks above are made of
# (tests/test_gpu_gald.py runs them against the reference's own module fixtures, g13_gald_modules).
class HarDBlock(_Engine):
    """hardnet_68.py:86-160 (the plain variant HarDNet-68 uses: keepBase False, no depthwise layers)."""
    RUN = _GaldRun

    def __init__(self, in_channels, growth_rate, grmul, n_layers, keepBase=False, residual_out=False, dwconv=False):
        super().__init__()
        if keepBase or dwconv:
            raise NotImplementedError("HarDNet-68 builds HarDBlock(keepBase=False, dwconv=False)")
        self._arg = _hard_block_units("", in_channels, growth_rate, grmul, n_layers)
        self.out_channels = self._arg[2]
        self._register(self._arg[0])

    def get_out_ch(self):
        return self.out_channels

    def _graph(self, run, x):
        return [_hard_block(run, x, *self._arg)]


class FAM(_Engine):
    """gcpa_gald.py:47-107: forward(left, down, right)."""
    RUN = _GaldRun

    def __init__(self, in_channel_left, in_channel_down, in_channel_right, interplanes=256):
        super().__init__()
        self._U, flat = _fam_units("", in_channel_left, in_channel_down, in_channel_right, interplanes)
        self._c = interplanes
        self._register(flat)

    def _graph(self, run, left, down, right):
        return [_fam_block(run, self._U, self._c, left, down, right)]


class CrissCrossAttention(_Engine):
    """contextagg/ccnet.py:37-127.  `recurrence` (default 1, not a constructor argument of the reference) applies the module that many times inside
    ONE graph with shared parameters - what GCPADecoder does with its `long_relation` (gcpa_cc2.py:56-57)."""
    RUN = _GaldRun
    recurrence = 1

    def __init__(self, in_dim):
        super().__init__()
        order = _cca_units("", in_dim)
        self._cca = order[1:]
        self._register(order)

    def _graph(self, run, x):
        for _ in range(self.recurrence):
            x = run.criss_cross(x, *self._cca, self.gamma)
        return [x]


class LocalAttenModule(_Engine):
    """contextagg/GALDNet.py:124-157."""
    RUN = _GaldRun

    def __init__(self, inplane):
        super().__init__()
        self._units = _lam_units("", inplane)
        self._lam = list(self._units)
        self._register(self._lam)

    def _graph(self, run, x):
        return [_local_attention(run, x, self._lam)]


class _NhwcCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_nchw, labels, ignore_index, holder):
        nhwc = logits_nchw.permute(0, 2, 3, 1)                                     # the decoder's outputs ARE NHWC memory: a free view
        if not nhwc.is_contiguous():
            nhwc = nhwc.contiguous()
        out, d = gk.gce(nhwc, labels.contiguous(), ignore_index, want_grad=logits_nchw.requires_grad)
        if holder is not None:
            _count_bad_labels(holder, out)
        ctx.d = d
        return out[0].clone()

    @staticmethod
    def backward(ctx, gout):
        d = ctx.d
        ctx.d = None
        return (d * gout).permute(0, 3, 1, 2), None, None, None


class CrossEntropyNHWC(nn.Module):
    """torch.nn.CrossEntropyLoss(ignore_index=255) of gald_trainer.py:107 on the HIP kernel (logits [B,K,H,W] fp32, labels [B,H,W] int64)."""

    def __init__(self, ignore_index=255):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, logits, target):
        if not logits.is_cuda:
            raise _lib.MiError("CrossEntropyNHWC runs on the MI355X only")
        return _NhwcCEFn.apply(logits.float(), target.long(), self.ignore_index, self)


class GALDTrainer(BaseTrainer):
    """gald_trainer.py:13-112: Adam for the encoder (BASE_LR) and the decoder (10 x), poly learning rate per iteration, four cross-entropies
    weighted 1 / 0.8 / 0.6 / 0.4 (out2 .. out5), checkpoint {'epoch', 'iteration', 'encoder', 'decoder', 'optimizer_enc', 'optimizer_dec'} as
    Gald-<epoch>.pth."""

    def __init__(self, name, cfg, train_loader, local_rank, logger=None):
        super().__init__(name, cfg, train_loader, local_rank, logger)

    def init_params(self):
        self.encoder = GCPAEncoder().to(self.device)
        self.decoder = GCPADecoder(self.cfg.MODEL.NUM_CLASSES).to(self.device)
        self.encoder.ensure_flat()
        self.decoder.ensure_flat()
        self.optimizer_enc = FlatAdam(self.encoder, self.cfg.SOLVER.BASE_LR)
        self.optimizer_dec = FlatAdam(self.decoder, self.cfg.SOLVER.BASE_LR * 10)
        self.iteration = 0
        self.criterion = CrossEntropyNHWC(ignore_index=255)

    def _save_checkpoint(self, epoch, save_path):
        torch.save({"epoch": epoch, "iteration": self.iteration, "encoder": self.encoder.state_dict(), "decoder": self.decoder.state_dict(),
                    "optimizer_enc": self.optimizer_enc.state_dict(), "optimizer_dec": self.optimizer_dec.state_dict()}, save_path)

    def _load_checkpoint(self):
        self.checkpoint = torch.load(self.cfg.resume, map_location=self.device)
        self.encoder.load_state_dict(self.checkpoint["encoder"])
        self.decoder.load_state_dict(self.checkpoint["decoder"])
        for key, opt in (("optimizer_enc", self.optimizer_enc), ("optimizer_dec", self.optimizer_dec)):
            if key in self.checkpoint:
                self.logger.info("Loading %s from %s" % (key.replace("_", " "), self.cfg.resume))
                opt.load_state_dict(self.checkpoint[key])
                opt._m = None
                opt._ensure_moments()
        if "iteration" in self.checkpoint:
            self.iteration = self.checkpoint["iteration"]
        if "epoch" in self.checkpoint:
            self.start_epoch = self.checkpoint["epoch"] + 1

    def train_step(self, src_input, src_label, max_iter):
        """gald_trainer.py:54-90 for one minibatch; returns (weighted loss, learning rate)."""
        lr = adjust_learning_rate(self.cfg.SOLVER.LR_METHOD, self.cfg.SOLVER.BASE_LR, self.iteration, max_iter, power=self.cfg.SOLVER.LR_POWER)
        for g in self.optimizer_enc.param_groups:
            g["lr"] = lr
        for g in self.optimizer_dec.param_groups:
            g["lr"] = lr * 10
        self.optimizer_enc.zero_grad()
        self.optimizer_dec.zero_grad()
        src_input = src_input.to(self.device, non_blocking=True)
        src_label = src_label.to(self.device, non_blocking=True).long()
        loss5, loss4, loss3, loss2 = self.decoder.losses(src_input, self.encoder(src_input), src_label)          # = criterion(out_i, label), fused
        loss = loss2 * 1 + loss3 * 0.8 + loss4 * 0.6 + loss5 * 0.4
        loss.backward()
        self.optimizer_enc.step()
        self.optimizer_dec.step()
        self.iteration += 1
        return loss.detach(), lr

    def _train_epoch(self, epoch):
        n = len(self.train_loader)
        max_iter = self.cfg.SOLVER.EPOCHS * n
        for i, (src_input, src_label, _) in enumerate(self.train_loader):
            loss, _ = self.train_step(src_input, src_label, max_iter)
            self.lr_data.append(self.optimizer_enc.param_groups[0]["lr"])
            self.loss_data.append(loss.item())
            bad = take_bad_labels(self.decoder, self.criterion)      # (the loss was fetched above: the step is complete on the device)
            if bad:
                raise ValueError("train labels: %d label values lie outside [0, %d) and are not ignore_index - torch.nn.CrossEntropyLoss "
                                 "(gald_trainer.py:107) would raise a device assert; map the label ids to train ids first" % (bad, self.cfg.MODEL.NUM_CLASSES))
            if i % 20 == 0 or i == n:
                self.logger.info("{} Epoch [{:03d}/{:03d}], Step [{:04d}/{:04d}], loss: [{:0.4f}], encode_learning_rate: [{:0.8f}], decode_learning_rate: [{:0.8f}]".format(
                    datetime.now(), epoch, self.cfg.SOLVER.EPOCHS, i, n, self.loss_data[-1], self.optimizer_enc.param_groups[0]["lr"], self.optimizer_dec.param_groups[0]["lr"]))
        save_path = self.cfg.OUTPUT_DIR
        os.makedirs(save_path, exist_ok=True)
        if epoch % self.cfg.SOLVER.CHECKPOINT_PERIOD == 0:
            self._save_checkpoint(epoch, os.path.join(save_path, "Gald-%d.pth" % epoch))
            self.logger.info("[Saving Snapshot:] " + save_path + "Gald-{}.pth".format(epoch))

    def _val_epoch(self, epoch):
        raise NotImplementedError("the reference's GALDTrainer has no validation epoch")

    def train(self):
        self.iteration = (self.start_epoch - 1) * len(self.train_loader)
        self.logger.info("#" * 20 + " Start Gald Training " + "#" * 20)
        self.encoder.train()
        self.decoder.train()
        for epoch in range(self.start_epoch, self.cfg.SOLVER.EPOCHS + 1):
            self._train_epoch(epoch)
        if self.local_rank == 0:
            dump_json(os.path.join(self.cfg.OUTPUT_DIR, "gald_chart_params.json"), {"learning rate": self.lr_data, "loss": self.loss_data})


class GALDTester:
    """core/testers/gald_tester.py:10-90, device-agnostic and runnable: encoder + decoder in eval(), res2 resized to the label size
    (align_corners False), softmax, argmax, confusion matrix + intersection / union meters, summary, gald_confusion_matrix.json.  Upstream the loop
    cannot run as written - `cmt` is used before assignment (:77) and `self.trainid2name` is never set (:87): here the matrix starts as zeros
    [K, K] like aspp_tester.py:53 and `trainid2name` is an optional constructor argument (class indices as names when absent)."""

    def __init__(self, cfg, device, test_loader, logger, palette, saveres=False, trainid2name=None):
        self.cfg, self.logger, self.test_loader, self.device = cfg, logger, test_loader, device
        self.palette, self.saveres, self.trainid2name = palette, saveres, trainid2name
        self.encoder = GCPAEncoder()
        self.decoder = GCPADecoder(cfg.MODEL.NUM_CLASSES) if cfg.MODEL.NUM_CLASSES != 19 else GCPADecoder()
        self.encoder.to(device)
        self.decoder.to(device)
        precision = cfg.TEST.PRECISION if "PRECISION" in cfg.TEST else "fp32"      # 'fp32': the reference's precision (csrc/gf32.hip); 'bf16': the training regime
        self.encoder.set_precision(precision)
        self.decoder.set_precision(precision)

    def _load_checkpoint(self):
        self.logger.info("Loading checkpoint from {}".format(self.cfg.resume))
        checkpoint = torch.load(self.cfg.resume, map_location=self.device)
        self.encoder.load_state_dict(checkpoint["encoder"])
        self.decoder.load_state_dict(checkpoint["decoder"])

    def save_distill(self, output, name):
        """gald_tester.py:30-43: palette PNG of the argmax mask under PSEUDO_DIR/inference/<dataset>."""
        from PIL import Image
        folder = os.path.join(self.cfg.PSEUDO_DIR, "inference", self.cfg.DATASETS.TEST)
        os.makedirs(folder, exist_ok=True)
        mask = Image.fromarray(output.cpu().numpy().squeeze().argmax(0).astype("uint8"))
        mask.putpalette(list(self.palette))
        mask.save(os.path.join(folder, name[0] + ".png"))

    def predict(self, x, hw):
        """softmax(upsample(res2, label size)) [B,K,h,w] fp32 (gald_tester.py:59-69)."""
        with torch.no_grad():
            res2 = self.decoder(x, self.encoder(x))[3]
            nhwc = res2.float().permute(0, 2, 3, 1).contiguous()
            if tuple(nhwc.shape[1:3]) != tuple(hw):
                nhwc = gk.gresize(nhwc, hw, False)
            return torch.softmax(nhwc, dim=3).permute(0, 3, 1, 2)

    def test(self):
        num_classes = self.cfg.MODEL.NUM_CLASSES
        self.encoder.eval()
        self.decoder.eval()
        self.meter = AverageMeter()
        cmt = torch.zeros(num_classes, num_classes, dtype=torch.int64)
        for x, y, name in self.test_loader:
            x = x.to(self.device, non_blocking=True)
            y = y.to(self.device, non_blocking=True).long()
            output = self.predict(x, tuple(y.shape[-2:]))
            pred = output.max(1)[1]
            if self.saveres:
                self.save_distill(output, name)
            cmt = cmt + confusion_matrix(self.cfg, torch.flatten(pred), torch.flatten(y))
            inter, union, target, res = intersectionAndUnionGPU(pred, y, num_classes, self.cfg.INPUT.IGNORE_LABEL)
            self.meter.update(*[t.cpu().numpy() for t in (inter, union, target, res)])
        self.meter.summary(self.logger, num_classes)
        names = list(self.trainid2name.values()) if self.trainid2name else [str(i) for i in range(num_classes)]
        os.makedirs(self.cfg.OUTPUT_DIR, exist_ok=True)
        dump_json(os.path.join(self.cfg.OUTPUT_DIR, "gald_confusion_matrix.json"), {"cmt": cmt.tolist(), "classes": names})
        return cmt
