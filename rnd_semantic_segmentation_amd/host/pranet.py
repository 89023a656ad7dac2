"""The PraNet path on the MI355X engine (BASELINE config[3]): Res2Net-50 v1b (26w x 4s) trunk, three RFB blocks, the
partial decoder and three reverse-attention branches, forward AND backward as a schedule of C-ABI launches (csrc/gconv.hip, gnet.hip).

  PraNet            reference core/models/classifiers/pranet/PraNet_Res2Net.py:98-179 (same state_dict keys: 922, same four outputs)
  trunk             reference core/models/classifiers/pranet/Res2Net_v1b.py:15-170
  structure_loss    reference core/trainers/pranet_trainer.py:22-31
  PraNetTrainer     reference core/trainers/pranet_trainer.py:12-104
  PranetTester      reference core/testers/pranet_tester.py:10-53

Design.  Activations are NHWC bf16; `torch.split` / `torch.cat` of the reference are channel-slice views of one buffer (a conv reads its
26-channel group in place and its BatchNorm writes straight into the concatenation the next conv reads).  Every BatchNorm2d runs on
batch statistics in train(): the conv's epilogue emits per-tile sums, `mi_gbn_finalize` turns them into mean / invstd / folded affine
and updates the running statistics, `mi_gbn_apply` normalises (+ ReLU, + residual).  The backward pass is a tape of closures recorded
by the forward, replayed in reverse: BatchNorm backward sums + apply, weight gradient into the parameter's slot of ONE flat fp32
gradient buffer (clamped Adam updates it in one launch), data gradient.  The one-channel side maps are fp32.  No tensor visits the CPU.
"""
import math
import os
from datetime import datetime

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .. import gk
from .. import kernels as K
from . import arch
from .engine import FlatStore, _SideStream, _off_path
from .plugin import BaseTrainer


class StructureLossFn(torch.autograd.Function):
    """loss = structure_loss(pred, mask) of pranet_trainer.py:22-31 (weighted IoU + the batch-mean BCE the reference's `reduce='none'`
    actually computes).  pred [B,1,H,W] fp32 (logits), mask [B,1,H,W] fp32 in [0,1]; the gradient flows to pred only."""

    @staticmethod
    def forward(ctx, pred, mask):
        loss, grad = K.structure_loss(pred.contiguous(), mask.contiguous(), want_grad=pred.requires_grad)
        ctx.save_for_backward(grad)
        return loss.clone()

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return (grad * gout if grad is not None else None), None


def structure_loss(pred, mask):
    return StructureLossFn.apply(pred.float(), mask.float())


# ------------------------------------------------------------------------------------------------ architecture table
class _Unit:
    """One conv (+ its BatchNorm2d): parameter handles, geometry (kh, kw, sh, sw, ph, pw, dh, dw), packed-operand views."""
    __slots__ = ("key", "bnkey", "cin", "cout", "geom", "weight", "bn", "bias", "wp", "wpt", "depthwise")

    def __init__(self, key, bnkey, cin, cout, k, stride=1, pad=0, dil=1):
        kh, kw = (k, k) if isinstance(k, int) else k
        ph, pw = (pad, pad) if isinstance(pad, int) else pad
        self.key, self.bnkey, self.cin, self.cout = key, bnkey, cin, cout
        self.geom = (kh, kw, stride, stride, ph, pw, dil, dil)
        self.weight = self.bn = self.bias = self.wp = self.wpt = None
        self.depthwise = False          # bias: None, or True before registration = "this conv carries a bias" (set by the architecture tables)


def _res2net_units(layers=(3, 4, 6, 3), base_width=26, scale=4):
    """Units of the Res2Net v1b trunk in the reference's registration order (= state_dict order), and a per-block description."""
    units = [_Unit("resnet.conv1.0", "resnet.conv1.1", 3, 32, 3, 2, 1), _Unit("resnet.conv1.3", "resnet.conv1.4", 32, 32, 3, 1, 1),
             _Unit("resnet.conv1.6", "resnet.bn1", 32, 64, 3, 1, 1)]
    blocks = []
    inplanes = 64
    for li, (planes, n, stride) in enumerate(zip((64, 128, 256, 512), layers, (1, 2, 2, 2)), 1):
        width = int(math.floor(planes * (base_width / 64.0)))
        for b in range(n):
            name = "resnet.layer%d.%d" % (li, b)
            s = stride if b == 0 else 1
            blk = dict(name=name, width=width, stride=s, stage=b == 0, down=None)
            blk["conv1"] = _Unit(name + ".conv1", name + ".bn1", inplanes, width * scale, 1)
            blk["convs"] = [_Unit("%s.convs.%d" % (name, i), "%s.bns.%d" % (name, i), width, width, 3, s, 1) for i in range(scale - 1)]
            blk["conv3"] = _Unit(name + ".conv3", name + ".bn3", width * scale, planes * 4, 1)
            order = [blk["conv1"]] + blk["convs"] + [blk["conv3"]]
            if b == 0 and (stride != 1 or inplanes != planes * 4):
                blk["down"] = _Unit(name + ".downsample.1", name + ".downsample.2", inplanes, planes * 4, 1)
                order.append(blk["down"])
            units += order
            blocks.append(blk)
            inplanes = planes * 4
    return units, blocks


def _rfb_units(name, cin, c):
    u = {"b0": [_Unit(name + ".branch0.0", None, cin, c, 1)]}
    for i, k in ((1, 3), (2, 5), (3, 7)):
        p = "%s.branch%d" % (name, i)
        u["b%d" % i] = [_Unit(p + ".0", None, cin, c, 1), _Unit(p + ".1", None, c, c, (1, k), 1, (0, k // 2)), _Unit(p + ".2", None, c, c, (k, 1), 1, (k // 2, 0)),
                        _Unit(p + ".3", None, c, c, 3, 1, k, k)]
    u["cat"] = _Unit(name + ".conv_cat", None, 4 * c, c, 3, 1, 1)
    u["res"] = _Unit(name + ".conv_res", None, cin, c, 1)
    flat = u["b0"] + u["b1"] + u["b2"] + u["b3"] + [u["cat"], u["res"]]
    for x in flat:                                   # BasicConv2d: <name>.conv.weight, <name>.bn.*
        x.bnkey = x.key + ".bn"
        x.key = x.key + ".conv"
    return u, flat


def _basic(name, cin, cout, k, pad=0):
    return _Unit(name + ".conv", name + ".bn", cin, cout, k, 1, pad)


_SIDE_MIN_WORK = 8e9          # weight gradients at least this large go to the side stream one by one (GALD: flat from 2 to 16 GFLOP; none: -3 %)
_WQ_BUDGET = int(float(os.environ.get("MI_WGRAD_QUEUE_MB", "2048")) * (1 << 20))
# With a side stream (GALD) the queue is flushed every few convs, so that the table-driven launches run beside the data-gradient chain instead of alone at the end
# of the tape: 4 jobs per launch 176.8 images/s, 8: 175.9, 2: 175.3, 16: 174.0, only at the end: 172.1 (one box, two rounds)
_WQ_SIDE_JOBS = int(os.environ.get("MI_TAPE_WQ_JOBS", "4"))


# ------------------------------------------------------------------------------------------------ tape
class _Var:
    """A tensor of the schedule with its gradient slot.  `own`: the gradient tensor belongs to this variable alone (in-place accumulation is
    safe); `want`: where the gradient should be assembled (a channel slice of the parent's gradient buffer)."""
    __slots__ = ("t", "g", "own", "want", "needs")

    def __init__(self, t, needs=True):
        self.t, self.g, self.own, self.want, self.needs = t, None, False, None, needs


def _acc(v, t, own):
    if not v.needs:
        return
    if v.g is None:
        if v.want is not None:
            if t.data_ptr() != v.want.data_ptr():
                gk.gbinary(gk.OP_COPY, t, out=v.want)
            v.g, v.own = v.want, True
        else:
            v.g, v.own = t, own
    elif v.own:
        gk.gbinary(gk.OP_ADD, v.g, t, out=v.g)
    else:
        v.g, v.own = gk.gbinary(gk.OP_ADD, v.g, t), True


def _rup32(c):
    return (c + 31) // 32 * 32


_TILE_MIN_PIXELS = 16384          # below this the MFMA-tile kernels do not fill the chip (4096 measured: GALD 169.1 vs 170.1, PraNet 1040 vs 1046 images/s)


def _tile_route(u, pixels):
    """Geometry half of _mfma_tile_ok: a conv whose shape the implicit-GEMM kernels of the DeepLab path can take (csrc/igemm_nt.hip / igemm_pp.hip /
    igemm_tn.hip: 128 .. 320-row MFMA tiles, LDS-DMA staging, 4x the throughput of the general kernel on large shapes): square 1x1 / 3x3 taps with one
    stride / padding / dilation, enough pixels to fill the chip, and channel counts that are either 64-multiples on both sides or - stride 1 - pad to
    32-multiples with less than 1.6x the work (HarDNet's gathered layers, 466 -> 168 as 480 -> 192: the general kernel's packs are zero-padded to 32 on
    both sides, i.e. they ARE the [taps][N][Ca] operands of those kernels for the padded shape; round 5) - unless the padded Cin is no 64-multiple AND the
    layer has fewer than 192 output columns: such a launch can only take the 256-column main loop and would leave most of it empty (152 -> 58, 218 -> 78)."""
    kh, kw, sh, sw, ph, pw, dh, dw = u.geom
    if u.depthwise or kh != kw or kh not in (1, 3) or sh != sw or ph != pw or dh != dw or pixels < _TILE_MIN_PIXELS:
        return False
    if u.cin % 64 == 0 and u.cout % 64 == 0:
        return True
    ci, co = _rup32(u.cin), _rup32(u.cout)
    work = 2.0 * pixels * u.cin * u.cout * kh * kw              # the small ones stay where they are: nothing to win on a 5 GFLOP launch
    if ci % 64 and co < 192:      # a padded Cin that only the 256-column main loop takes (mi_conv_gemm: Ca % 64 != 0), with too few output columns to fill it
        return False
    return sh == 1 and 2 * ph == dh * (kh - 1) and work >= 8e9 and ci * co < 1.6 * u.cin * u.cout and os.environ.get("MI_TILE_PAD", "1") != "0"


def _mfma_tile_ok(u, x, out=None, out_f32=False):
    """The conv goes to the MFMA-tile kernels: _tile_route() and an input that IS the kernels' operand - a contiguous NHWC tensor of the 32-padded channel
    count (64-multiples: the tensor itself; otherwise a gather buffer its producer allocated padded, pad channels zero: gald._hard_block)."""
    if out_f32 or not _tile_route(u, x.shape[0] * x.shape[1] * x.shape[2]) or x.shape[-1] != _rup32(u.cin):
        return False
    return x.is_contiguous() and (out is None or (out.is_contiguous() and out.shape[-1] == _rup32(u.cout)))


def _conv_forward(x, u, bias, stats, out=None, out_f32=False, net=None):
    """(y, statistics partials or None): the general kernel, or the MFMA-tile kernels with the BatchNorm sums out of their epilogue (mi_conv_gemm_stats;
    one extra pass of column sums where a bias or a slot output rules that entry out).  With padded channel counts y is the [.., :cout] view of the
    kernels' 32-padded output (pad columns: products with the pack's zero rows)."""
    if not _mfma_tile_ok(u, x, out, out_f32):
        return gk.gconv(x, u.wp, u.cout, u.geom, out=out, bias=bias, stats=stats, out_f32=out_f32)
    k, s, p, d = u.geom[0], u.geom[2], u.geom[4], u.geom[6]
    hw = gk.conv_out_hw(x.shape[1], x.shape[2], *u.geom)
    np_, cp = _rup32(u.cout), _rup32(u.cin)
    wp = u.wp.view(k * k, np_, cp)
    cut = (lambda t: t) if np_ == u.cout else (lambda t: t[..., :u.cout])
    if stats and bias is None and out is None:          # sum y and sum y^2 out of the conv's own epilogue (pilot 0: raw sums)
        y, sums, _ = K.conv_gemm_stats(x, wp, hw, k, s, p, d, net._zeros(np_))
        return cut(y), (sums if np_ == u.cout else sums[:, :u.cout].contiguous()).view(-1)
    if bias is not None and np_ != u.cout:
        bias = torch.cat([bias, bias.new_zeros(np_ - u.cout)])
    y = cut(K.conv_gemm(x, wp, hw, k, s, p, d, K.GATHER_FWD, scale=None if bias is None else net._ones(np_), bias=bias, out=out))
    st = None
    if stats:                       # sum y and sum y^2 in one pass: the backward-sums kernel with g = y, mean = 0, invstd = 1
        st = torch.empty((2, u.cout), dtype=torch.float32, device=x.device)
        gk.gbn_bwd_sums(y, y, None, net._zeros(u.cout), net._ones(u.cout), st[0], st[1])
        st = st.view(-1)
    return y, st


def _grad_target(v):
    """Where a kernel may write d loss / d v directly: the assembly slot if there is one and nothing has been written yet."""
    return v.want if (v.want is not None and v.g is None) else None


class _Run:
    """One forward pass.  train: BatchNorm2d on batch statistics (module.training); rec: record the backward tape."""
    WGRAD_STREAM = False          # the large weight gradients of backward() on the side stream (see _conv_backward)

    def __init__(self, net, train, rec):
        self.net, self.train, self.rec, self.tape, self.side = net, train, rec, [], None
        self.wq, self.wq_slots, self.wq_fix = None, None, []            # weight gradients queued during backward (see _conv_backward)
        # MI_APPLY_MULTI=0: every gather copy / hierarchical add as its own launch again (the BatchNorm apply then has one destination; same bits)
        self.multi = os.environ.get("MI_APPLY_MULTI", "1") != "0"
        self.wq_bytes = 0
        # fp32: the evaluation forward in the reference's precision (csrc/gf32.hip; _Engine.set_precision): every activation fp32, every conv with
        # its eval()-BatchNorm affine, residual and activation in one launch
        self.f32 = (not train) and getattr(net, "precision", "bf16") == "fp32"
        if self.f32 and rec:
            raise _lib.MiError("precision 'fp32' is the evaluation forward (no backward kernels exist in fp32): run under torch.no_grad(), or set_precision('bf16')")

    def record(self, fn):
        if self.rec:
            self.tape.append(fn)

    def var(self, t, needs=True):
        return _Var(t, needs and self.rec)

    def tap(self, name, v):
        """Named intermediates.  A module with a `_taps` dict keeps them (tools/dbg, tests).  Teacher forcing (tests): a module with a `_force` dict
        has the named activations REPLACED in place - after the engine's own value went to `_taps` - by the given tensors, and one with a
        `_force_grad` dict has d loss / d (that activation) replaced - after the engine's own accumulated gradient went to `_gtaps` - before the
        producer's backward runs.  The tests feed every block the oracle's activation and upstream gradient and compare what the block makes of
        them with the oracle's next activation / gradients: an error is attributed to the block that makes it instead of being amplified through
        the rest of the net.  Forced tensors are NCHW (any float dtype, on the module's device)."""
        net = self.net
        taps, force = getattr(net, "_taps", None), getattr(net, "_force", None)
        if force is not None and name in force:
            if taps is not None:
                taps[name] = _Var(v.t.clone(), False)
            f = force[name]
            v.t.copy_(f.permute(0, 2, 3, 1) if f.dim() == 4 else f)
        elif taps is not None:
            taps[name] = v
        gtaps, fgrad = getattr(net, "_gtaps", None), getattr(net, "_force_grad", None)
        if self.rec and (gtaps is not None or fgrad is not None):
            def back():                 # runs after every consumer's backward and before the producer's
                if gtaps is not None and v.g is not None:
                    gtaps[name] = v.g.clone()
                if fgrad is not None and name in fgrad:
                    f = fgrad[name]
                    f = (f.permute(0, 2, 3, 1) if f.dim() == 4 else f).to(v.t.dtype)
                    if v.want is not None:
                        v.want.copy_(f)
                        v.g, v.own = v.want, True
                    else:
                        v.g, v.own = f.contiguous(), True
            self.record(back)
        return v

    # ---- conv (+ bias) + BatchNorm2d (+ add) (+ ReLU | ReLU6): BasicConv2d of PraNet_Res2Net.py:7-20, the conv/bn pairs of Res2Net_v1b.py,
    #      ConvLayer of hardnet_68.py:56-80 (relu=6), the conv(bias)-bn-relu stems of FAM (gcpa_gald.py:84-86)
    def _apply(self, y, sc, sh, act, add, out, out_f32, extras):
        """BatchNorm apply; `extras` = [(c0, c1, dst, add2 _Var or None), ...]: channel ranges of the result that also go elsewhere in the same launch."""
        if not extras:
            return gk.gbn_apply(y, sc, sh, act, add=add, out=out, out_f32=out_f32)
        return gk.gbn_apply_multi(y, sc, sh, act, [(c0, c1, d, None if a2 is None else a2.t) for c0, c1, d, a2 in extras], add=add, out=out)

    def conv_bn(self, x, u, relu, add=None, out=None, out_f32=False, extras=None):
        net, bn = self.net, u.bn
        act = 2 if relu == 6 else int(bool(relu))
        bias = None if u.bias is None else u.bias.detach()
        if extras and out_f32:
            raise _lib.MiError("conv_bn: extra destinations go with a bf16 output")
        if not self.train:
            sc, sh = net._eval_fold(u)
            if self.f32:
                o = gk.gconv_f32(x.t, u.weight.detach(), u.geom, bias=bias, scale=sc, shift=sh, add=None if add is None else add.t, relu=relu, out=out)
                for c0, c1, d, a2 in (extras or ()):          # fp32 evaluation: the extra destinations as their own element-wise launches
                    gk.gbinary(gk.OP_COPY, o[..., c0:c1], out=d) if a2 is None else gk.gbinary(gk.OP_ADD, o[..., c0:c1], a2.t, out=d)
                return self.var(o, False)
            y, _ = _conv_forward(x.t, u, bias, False, net=net)
            return self.var(self._apply(y, sc, sh, act, None if add is None else add.t, out, out_f32, extras), False)
        hw = gk.conv_out_hw(x.t.shape[1], x.t.shape[2], *u.geom)
        if not _mfma_tile_ok(u, x.t) and gk.gconv_bn_fits(x.t.shape[0], *hw) and os.environ.get("MI_BN_INLAUNCH", "0") == "1":
            # small maps, opt-in (MI_BN_INLAUNCH=1): the conv's last workgroup finalizes the statistics itself (one launch instead of two; the same bits).
            # Off by default since round 5: under the HIP-graph replay PraNetTrainer runs it costs 1 % (1 035 vs 1 045 images/s, profiles/r05_inlaunch_ab.txt)
            y, fin = gk.gconv_bn(x.t, u.wp, u.cout, u.geom, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, bias=bias)
            M = y.shape[0] * y.shape[1] * y.shape[2]
        else:
            y, st = _conv_forward(x.t, u, bias, True, net=net)
            M = y.shape[0] * y.shape[1] * y.shape[2]
            fin = gk.gbn_finalize(st, u.cout, M, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)      # mean, invstd, scale, shift
        o = self._apply(y, fin[2], fin[3], act, None if add is None else add.t, out, out_f32, extras)
        ov = self.var(o)

        def back():
            g = ov.g
            ov.g = None
            if g is None:
                return
            mask = None
            if add is not None:
                if act:
                    g = gk.gbinary(gk.OP_RELU_MASK, g, o)
                _acc(add, g, False)
            elif act:
                mask = o
            (dbeta, a1), (dgamma, a2) = net._grad_slot(bn.bias), net._grad_slot(bn.weight)
            if a1 != a2:
                raise _lib.MiError("BatchNorm weight / bias gradient slots out of step")
            gk.gbn_bwd_sums(g, y, mask, fin[0], fin[1], dbeta, dgamma, accumulate=a1, relu6=act == 2)
            if a1:          # a module applied twice (shared parameters): this application's own sums, not the accumulated ones, enter its dy
                db1, dg1 = torch.empty_like(dbeta), torch.empty_like(dgamma)
                gk.gbn_bwd_sums(g, y, mask, fin[0], fin[1], db1, dg1, relu6=act == 2)
            else:
                db1, dg1 = dbeta, dgamma
            dyp = None
            if u.cout % 32 and _mfma_tile_ok(u, x.t):
                # the MFMA-tile kernels contract over the 32-padded channel count: d loss / d y goes into the real columns of a padded tensor, pads zero
                dyp = gk.new(y.shape[0], y.shape[1], y.shape[2], _rup32(u.cout), y.device)
                dyp[..., u.cout:].zero_()
            dy = gk.gbn_bwd_apply(g, y, mask, fin[0], fin[1], bn.weight, db1, dg1, M, out=None if dyp is None else dyp[..., :u.cout], relu6=act == 2)
            if u.bias is not None:
                slot, acc = net._grad_slot(u.bias)
                gk.gbn_bwd_sums(dy, None, None, None, None, slot, None, accumulate=acc)
            self._conv_backward(x, u, dy if dyp is None else dyp)
        self.record(back)
        return ov

    def _conv_backward(self, x, u, dy):
        """Weight gradient (off the critical path: nothing reads it before the optimizer, so it runs on the side stream beside the data-gradient
        chain - the convs of these nets are far too small to fill 256 CUs alone) and data gradient."""
        slot, acc = self.net._grad_slot(u.weight)
        # MI_TAPE_WGRAD_STREAM (default: the run class's WGRAD_STREAM - off for PraNet, on for GALD).  Measured: every weight gradient on the side stream costs
        # PraNet 6 % as a graph and 18 % eager (hundreds of 20-60 us launches, each fork / join a dependency the GPU has to resolve); only the launches of
        # >= 8 GFLOP there: PraNet still -10 % as a graph (659 vs 734 images/s: a second stream in the capture changes how the whole graph is scheduled),
        # GALD (eager, its decoder's and padded gathers' weight gradients are 100 - 400 us launches) +1.5 % (175.6 vs 173.0 images/s, round 5)
        work = 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * u.cout * (1 if u.depthwise else u.cin) * u.geom[0] * u.geom[1]
        side = self.side if work >= _SIDE_MIN_WORK else None
        if _mfma_tile_ok(u, x.t) and dy.is_contiguous() and dy.shape[-1] == _rup32(u.cout):
            k, s, p, d = u.geom[0], u.geom[2], u.geom[4], u.geom[6]
            np_, cp = _rup32(u.cout), _rup32(u.cin)
            def wgrad(dw, accumulate):
                # on the side stream the launch runs beside the data-gradient chain: the deferred-reducer form plans its split for that (mi_conv_wgrad_partial)
                if side is not None:                     # (GALD: 176.6 vs 175.9 images/s)
                    b = K.WgradBatch()
                    K.conv_wgrad(dy, x.t, dw, k, s, p, d, accumulate=accumulate, batch=b)
                    b.flush()
                else:
                    K.conv_wgrad(dy, x.t, dw, k, s, p, d, accumulate=accumulate)
            if np_ == u.cout and cp == u.cin:
                _off_path(side, lambda: wgrad(slot, acc), dy, x.t)
            else:           # padded operands: the gradient of the padded weight, its real corner into the parameter's slot
                def padded_wgrad():
                    wide = torch.empty((np_, cp, k, k), dtype=torch.float32, device=dy.device)
                    wgrad(wide, False)
                    slot.add_(wide[:u.cout, :u.cin]) if acc else slot.copy_(wide[:u.cout, :u.cin])
                _off_path(side, padded_wgrad, dy, x.t)
            if x.needs:
                tgt = _grad_target(x)
                dx = K.conv_gemm(dy, u.wpt.view(k * k, cp, np_), (x.t.shape[1], x.t.shape[2]), k, s, p, d, K.GATHER_DGRAD,
                                 out=tgt if (tgt is not None and tgt.is_contiguous() and tgt.shape[-1] == cp) else None)
                _acc(x, dx, True)
            return
        if x.t.shape[-1] != u.cin and not u.depthwise:
            # the zero-padded image (_nhwc_input): the gradient for its eight channels goes to a scratch tensor, the real channels are cut out after the flush
            wide = torch.empty((u.cout, x.t.shape[-1]) + tuple(u.geom[:2]), dtype=torch.float32, device=dy.device)
            if self.wq is not None and side is None:
                self.wq.append((dy, x.t, wide, u.geom, False))
                self.wq_fix.append((slot, wide, u.cin, acc))
            else:
                gk.gconv_wgrad(dy, x.t, wide, u.geom)
                slot.add_(wide[:, :u.cin]) if acc else slot.copy_(wide[:, :u.cin])
        elif self.wq is not None and side is None:
            # queued: the whole backward's weight gradients run as one table-driven launch at the end of the tape (gk.gconv_wgrad_multi) - alone each
            # is a 25 - 60 us latency chain of which 15 - 25 us are fixed.  The queue keeps dy and x alive until then; a slot that is already in the
            # queue (a module applied twice) flushes first, so that the accumulation order stays the tape's.
            if slot.data_ptr() in self.wq_slots:
                self.flush_wgrads()
            self.wq.append((dy, x.t, slot, u.geom, acc))
            self.wq_slots.add(slot.data_ptr())
            # the queue keeps every dy alive (and its flush sums a private split-K slab per job): bounded, so that backward's peak memory does not
            # grow with the depth of the net - MI_WGRAD_QUEUE_MB of queued gradients (default 2048: PraNet at 16 x 352 x 352 and GALD at 6 x 720 x 1280
            # never reach it; a flush costs one more pair of launches)
            self.wq_bytes += dy.numel() * dy.element_size()
            if self.wq_bytes > _WQ_BUDGET or (self.side is not None and len(self.wq) >= _WQ_SIDE_JOBS):
                self.flush_wgrads()
        else:
            _off_path(side, lambda: gk.gconv_wgrad(dy, x.t, slot, u.geom, accumulate=acc), dy, x.t)
        if x.needs:
            dx, _ = gk.gconv(dy, u.wpt, u.cin, u.geom, out=_grad_target(x), mode=gk.GATHER_DGRAD, out_hw=(x.t.shape[1], x.t.shape[2]))
            _acc(x, dx, True)

    def stem_tail(self, x, u):
        """conv1.6 -> bn1 -> ReLU -> MaxPool2d(3, 2, 1) (Res2Net_v1b.py:149-152): conv with tile statistics, then normalise + ReLU + max-pool
        in ONE pass (mi_stem_pool_fwd with the batch affine); backward: pooled gradient routed by the stored argmax, then BatchNorm backward."""
        net, bn = self.net, u.bn
        if not self.train:
            sc, sh = net._eval_fold(u)
            if self.f32:
                return self.var(gk.gpool_f32(gk.gconv_f32(x.t, u.weight.detach(), u.geom, scale=sc, shift=sh, relu=True), 3, 2, 1, 2), False)
            y, _ = gk.gconv(x.t, u.wp, u.cout, u.geom)
            return self.var(K.stem_pool_fwd(y, sc, sh)[0], False)
        y, st = gk.gconv(x.t, u.wp, u.cout, u.geom, stats=True)
        M = y.shape[0] * y.shape[1] * y.shape[2]
        fin = gk.gbn_finalize(st, u.cout, M, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
        pool, idx = K.stem_pool_fwd(y, fin[2].contiguous(), fin[3].contiguous())
        ov = self.var(pool)

        def back():
            g = K.stem_pool_bwd(ov.g.contiguous(), idx, net._ones(u.cout), (y.shape[1], y.shape[2]))      # d loss / d relu(bn(y)), already ReLU-masked
            ov.g = None
            dbeta, dgamma = net._grad_of(bn.bias), net._grad_of(bn.weight)
            gk.gbn_bwd_sums(g, y, None, fin[0], fin[1], dbeta, dgamma)
            dy = gk.gbn_bwd_apply(g, y, None, fin[0], fin[1], bn.weight, dbeta, dgamma, M)
            self._conv_backward(x, u, dy)
        self.record(back)
        return ov

    def conv_bias(self, x, u, out_f32=True):
        """nn.Conv2d with bias and no BatchNorm: the one-channel / class-logit heads in fp32 (agg1.conv5, PraNet_Res2Net.py:77; linear2..5,
        gcpa_cc2.py:37-40) or a bf16 feature conv (conv_d1 / conv_d2 / conv_l of FAM, gcpa_gald.py:66-74; the q / k / v projections of ccnet.py:43-51)."""
        if self.f32:
            return self.var(gk.gconv_f32(x.t, u.weight.detach(), u.geom, bias=u.bias.detach()), False)
        o, _ = _conv_forward(x.t, u, u.bias.detach(), False, out_f32=out_f32, net=self.net)
        ov = self.var(o)

        def back():
            g = ov.g
            ov.g = None
            if g is None:
                return
            slot, acc = self.net._grad_slot(u.bias)
            gk.gbn_bwd_sums(g, None, None, None, None, slot, None, accumulate=acc)
            self._conv_backward(x, u, g if g.dtype == torch.bfloat16 else gk.gbinary(gk.OP_COPY, g, out_dtype=torch.bfloat16))
        self.record(back)
        return ov

    def binary(self, op, a, b, out=None):
        ov = self.var(gk.gbinary(op, a.t, b.t, out=out))

        def back():
            g = ov.g
            ov.g = None
            if g is None:
                return
            if op == gk.OP_ADD:
                _acc(a, g, False)
                _acc(b, g, False)
            else:
                _acc(a, gk.gbinary(gk.OP_MUL, g, b.t, out=_grad_target(a)), True)
                _acc(b, gk.gbinary(gk.OP_MUL, g, a.t, out=_grad_target(b)), True)
        self.record(back)
        return ov

    def added(self, a, b, t):
        """The variable of t = a + b that a producer's apply has ALREADY written (conv_bn extras): binary(OP_ADD)'s place on the tape without its launch."""
        ov = self.var(t)

        def back():
            g = ov.g
            ov.g = None
            if g is not None:
                _acc(a, g, False)
                _acc(b, g, False)
        self.record(back)
        return ov

    def alias_into(self, a, out):
        """copy_into whose copy the producer of `a` has already made (conv_bn extras)."""
        ov = self.var(out)

        def back():
            if ov.g is not None:
                _acc(a, ov.g, False)
                ov.g = None
        self.record(back)
        return ov

    def copy_into(self, a, out):
        ov = self.var(gk.gbinary(gk.OP_COPY, a.t, out=out))

        def back():
            if ov.g is not None:
                _acc(a, ov.g, False)
                ov.g = None
        self.record(back)
        return ov

    def avgpool(self, x, k, stride, pad, include_pad, out=None):
        H, W = x.t.shape[1], x.t.shape[2]
        if include_pad:
            Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        else:
            Ho, Wo = -(-H // stride), -(-W // stride)
        if self.f32:
            return self.var(gk.gpool_f32(x.t, k, stride, pad, 0 if include_pad else 1, (Ho, Wo), out=out), False)
        ov = self.var(gk.gavgpool(x.t, k, stride, pad, include_pad, (Ho, Wo), out=out))

        def back():
            if ov.g is not None:
                _acc(x, gk.gavgpool_bwd(ov.g, (H, W), k, stride, pad, include_pad, dx=_grad_target(x)), True)
                ov.g = None
        self.record(back)
        return ov

    def resize(self, x, factor, align, size=None):
        """F.interpolate(x, scale_factor=factor) or, with `size`, F.interpolate(x, size=size) (factor ignored), mode='bilinear'."""
        H, W = x.t.shape[1], x.t.shape[2]
        if size is not None:
            factor = None
        out_hw = tuple(size) if size is not None else (int(math.floor(H * factor)), int(math.floor(W * factor)))
        ov = self.var(gk.gresize(x.t, out_hw, align, factor))

        def back():
            if ov.g is not None:
                _acc(x, gk.gresize_bwd(ov.g, (H, W), align, factor), True)
                ov.g = None
        self.record(back)
        return ov

    def reverse_attention(self, gate, feat):
        if self.f32:
            return self.var(gk.gpoint_f32(gk.PW_REVERSE, feat.t, gate.t), False)
        ov = self.var(gk.gra_fwd(gate.t, feat.t))

        def back():
            if ov.g is None:
                return
            dfeat, dgate = gk.gra_bwd(gate.t, feat.t, ov.g)
            ov.g = None
            _acc(feat, dfeat, True)
            _acc(gate, dgate, True)
        self.record(back)
        return ov

    def cat(self, buf, pieces):
        """`buf` already holds the pieces (their producers wrote into its channel slices); the gradient of the concatenation is handed to
        the pieces as slice views."""
        ov = self.var(buf)

        def back():
            if ov.g is None:
                return
            off = 0
            for p in pieces:
                c = p.t.shape[-1]
                _acc(p, ov.g[..., off:off + c], True)
                off += c
            ov.g = None
        self.record(back)
        return ov

    def split(self, parent, width, n):
        """torch.split as channel-slice views.  The slices' gradients are assembled side by side in one buffer that becomes the parent's
        gradient: `slots` (recorded by the caller AFTER the slices' consumers, so it runs before their backward) hands every slice its channel
        range; `gather` (recorded here, i.e. run after them) completes the buffer."""
        parts = [self.var(parent.t[..., i * width:(i + 1) * width]) for i in range(n)]
        state = {}

        def slots():
            B, H, W, C = parent.t.shape
            state["g"] = gk.new(B, H, W, C, parent.t.device)
            for i, p in enumerate(parts):
                p.want = state["g"][..., i * width:(i + 1) * width]

        def gather():
            for p in parts:
                if p.g is None:                                   # a slice nothing consumed
                    p.want.zero_()
                p.g = p.want = None
            _acc(parent, state.pop("g"), True)
        self.record(gather)
        return parts, slots

    def flush_wgrads(self):
        if self.wq:
            jobs, fix = self.wq, self.wq_fix

            def go():
                gk.gconv_wgrad_multi(jobs)
                for slot, wide, cin, acc in fix:
                    slot.add_(wide[:, :cin]) if acc else slot.copy_(wide[:, :cin])
            # with a side stream (GALD) the table-driven launch runs beside the data-gradient chain that is still being enqueued
            _off_path(self.side, go, *[t for j in jobs for t in j[:2]])
        if self.wq is not None:
            self.wq, self.wq_slots, self.wq_fix = [], set(), []
        self.wq_bytes = 0

    def backward(self):
        self.side = _SideStream.get(self.net._store.data.device) if os.environ.get("MI_TAPE_WGRAD_STREAM", "1" if self.WGRAD_STREAM else "0") == "1" else None
        self.wq, self.wq_slots = ([], set()) if os.environ.get("MI_WGRAD_BATCH", "1") != "0" else (None, None)
        self.wq_fix = []
        for fn in reversed(self.tape):
            fn()
        self.tape = []
        self.flush_wgrads()
        self.wq = None
        if self.side is not None:
            self.side.join()          # the caller (optimizer, gradient exchange) sees complete weight gradients on its own stream


# ------------------------------------------------------------------------------------------------ graph pieces
def _bottle2neck(run, x, blk):
    """Res2Net_v1b.py:63-92: 1x1 to four `width`-channel groups; groups 0..2 through 3x3 convs, each (in a 'normal' block) taking the
    previous group's output added to its own input; group 3 passes through ('normal') or through AvgPool2d(3, stride, 1) ('stage')."""
    w, s, stage = blk["width"], blk["stride"], blk["stage"]
    B, H, W = x.t.shape[0], x.t.shape[1], x.t.shape[2]
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    cat = gk.new(B, Ho, Wo, 4 * w, x.t.device, x.t.dtype)
    # 'normal' blocks (round 5): what used to be three element-wise launches comes out of the BatchNorm applies that produce the operands - conv1's apply also
    # writes the pass-through group into its slot of the concatenation, the apply of branch i also writes (its output + group i+1) = the next branch's input
    fuse = run.multi and not stage
    o1 = run.conv_bn(x, blk["conv1"], True, extras=[(3 * w, 4 * w, cat[..., 3 * w:], None)] if fuse else None)
    groups, slots = run.split(o1, w, 4)
    pieces, prev, sums, ahead = [], None, [], None
    for i in range(3):
        if i == 0 or stage:
            inp = groups[i]
        elif ahead is not None:
            inp = ahead
            sums.append((inp, groups[i]))
        else:
            inp = run.binary(gk.OP_ADD, prev, groups[i])
            sums.append((inp, groups[i]))
        nxt = gk.new(B, Ho, Wo, w, x.t.device, x.t.dtype) if (fuse and i < 2) else None
        prev = run.conv_bn(inp, blk["convs"][i], True, out=cat[..., i * w:(i + 1) * w], extras=None if nxt is None else [(0, w, nxt, groups[i + 1])])
        ahead = None if nxt is None else run.added(prev, groups[i + 1], nxt)
        pieces.append(prev)
    if stage:
        pieces.append(run.avgpool(groups[3], 3, s, 1, True, out=cat[..., 3 * w:]))
    else:
        pieces.append(run.alias_into(groups[3], cat[..., 3 * w:]) if fuse else run.copy_into(groups[3], cat[..., 3 * w:]))
    catv = run.cat(cat, pieces)
    if blk["down"] is not None:
        res = run.conv_bn(x if s == 1 else run.avgpool(x, s, s, 0, False), blk["down"], False)       # AvgPool2d(1, 1) is the identity
    else:
        res = x
    out = run.conv_bn(catv, blk["conv3"], True, add=res)

    def slots_and_sums():
        slots()
        for inp, grp in sums:          # the data gradient of convs[i] lands directly in group i's range: it IS d loss / d (prev + group i)
            inp.want = grp.want
    run.record(slots_and_sums)
    return out


def _rfb_block(run, x, units, c):
    """RFB_modified.forward (PraNet_Res2Net.py:50-59); BasicConv2d applies no ReLU (:17-20)."""
    B, H, W, _ = x.t.shape
    cat = gk.new(B, H, W, 4 * c, x.t.device, x.t.dtype)
    pieces = []
    for i in range(4):
        y = x
        chain = units["b%d" % i]
        for j, u in enumerate(chain):
            y = run.conv_bn(y, u, False, out=cat[..., i * c:(i + 1) * c] if j == len(chain) - 1 else None)
        pieces.append(y)
    xc = run.conv_bn(run.cat(cat, pieces), units["cat"], False)
    return run.conv_bn(x, units["res"], True, add=xc)                         # relu(x_cat + conv_res(x))


def _aggregation(run, a, c, x1, x2, x3):
    """aggregation.forward (PraNet_Res2Net.py:79-96); x1 coarsest."""
    up = lambda v: run.resize(v, 2, True)                                      # nn.Upsample(scale_factor=2, 'bilinear', align_corners=True)
    mul = lambda p, q, out=None: run.binary(gk.OP_MUL, p, q, out=out)
    B, H2, W2, _ = x2.t.shape
    _, H3, W3, _ = x3.t.shape
    cat2 = gk.new(B, H2, W2, 2 * c, x1.t.device, x1.t.dtype)
    cat3 = gk.new(B, H3, W3, 3 * c, x1.t.device, x1.t.dtype)
    up1 = up(x1)
    x2_1 = mul(run.conv_bn(up1, a["up1"], False), x2, out=cat2[..., :c])
    x3_1 = mul(mul(run.conv_bn(up(up1), a["up2"], False), run.conv_bn(up(x2), a["up3"], False)), x3, out=cat3[..., :c])
    p22 = run.conv_bn(up1, a["up4"], False, out=cat2[..., c:])
    x2_2 = run.conv_bn(run.cat(cat2, [x2_1, p22]), a["cat2"], False)
    p32 = run.conv_bn(up(x2_2), a["up5"], False, out=cat3[..., c:])
    x3_2 = run.conv_bn(run.cat(cat3, [x3_1, p32]), a["cat3"], False)
    return run.conv_bias(run.conv_bn(x3_2, a["conv4"], False), a["conv5"])


def _agg_units(prefix, c):
    a = prefix
    d = dict(up1=_basic(a + "conv_upsample1", c, c, 3, 1), up2=_basic(a + "conv_upsample2", c, c, 3, 1), up3=_basic(a + "conv_upsample3", c, c, 3, 1),
             up4=_basic(a + "conv_upsample4", c, c, 3, 1), up5=_basic(a + "conv_upsample5", 2 * c, 2 * c, 3, 1),
             cat2=_basic(a + "conv_concat2", 2 * c, 2 * c, 3, 1), cat3=_basic(a + "conv_concat3", 3 * c, 3 * c, 3, 1),
             conv4=_basic(a + "conv4", 3 * c, 3 * c, 3, 1), conv5=_Unit(a + "conv5", None, 3 * c, 1, 1))
    return d, [d[k] for k in ("up1", "up2", "up3", "up4", "up5", "cat2", "cat3", "conv4", "conv5")]


def _reverse_branch(run, gate, feat, units):
    """One reverse-attention branch (PraNet_Res2Net.py:130-140 / :145-153 / :158-166): erase what the coarser map marks, predict a residual."""
    y = run.conv_bn(run.reverse_attention(gate, feat), units[0], False)
    for u in units[1:-1]:
        y = run.conv_bn(y, u, True)                                             # F.relu(self.raX_convY(x))
    r = run.conv_bn(y, units[-1], False, out_f32=True)
    return run.binary(gk.OP_ADD, r, gate)


# ------------------------------------------------------------------------------------------------ the modules
class _Engine(nn.Module):
    """Parameter storage and launch preparation shared by the modules of this file: parameters registered under the reference's names,
    one flat fp32 buffer for them and one for their gradients (engine.FlatStore), every conv's bf16 operands packed by ONE table-driven
    launch when a weight changed, BatchNorm buffers as views of one buffer."""
    RUN = _Run          # the tape class a module's graph is written against (host/gald.py extends it)

    def _register(self, order):
        """order: _Unit objects and (key, tensor) pairs (parameters that belong to no conv: an unused classifier head, a scalar gate), in the
        reference's registration order (= state_dict order)."""
        self._units = []
        for u in order:
            if isinstance(u, tuple):
                key, value = u
                parent, leaf = key.rsplit(".", 1) if "." in key else ("", key)
                setattr(arch.node_at(self, parent) if parent else self, leaf, nn.Parameter(value))
                continue
            node = arch.node_at(self, u.key)
            kh, kw = u.geom[0], u.geom[1]
            w = torch.empty(u.cout, 1 if u.depthwise else u.cin, kh, kw)
            if u.key.startswith("resnet."):
                nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")          # Res2Net_v1b.py:113-115
            else:
                nn.init.kaiming_uniform_(w, a=math.sqrt(5))                              # nn.Conv2d default
            node.weight = nn.Parameter(w)
            u.weight = node.weight
            if u.bnkey is None or u.bias is True:      # a conv with bias (agg1.conv5; the FAM / local-attention convs in front of their BatchNorm)
                bound = 1.0 / math.sqrt((1 if u.depthwise else u.cin) * kh * kw)
                node.bias = nn.Parameter(torch.empty(u.cout).uniform_(-bound, bound))
                u.bias = node.bias
            if u.bnkey is not None:
                parent, leaf = u.bnkey.rsplit(".", 1) if "." in u.bnkey else ("", u.bnkey)
                (arch.node_at(self, parent) if parent else self).add_module(leaf, nn.BatchNorm2d(u.cout))
                u.bn = arch.node_at(self, u.bnkey)
            self._units.append(u)
        self._store = None
        self._pack_sig = None
        self._eval_cache = {}
        self._stat_flat = None
        self._stat_gen = 0
        self._ones_cache = {}

    def engine_parameters(self):
        return [(k, p) for k, p in self.named_parameters()]

    def ensure_flat(self):
        dev = self._units[0].weight.device
        if self._store is None or not self._store.intact() or self._store.data.device != dev:
            self._store = FlatStore(self.engine_parameters(), dev)
            self._pack_sig = None
            self._build_pack_plan(dev)
        if not self._buffers_intact(dev):
            self._flatten_buffers(dev)
        return self._store

    def _flatten_buffers(self, dev):
        """running_mean / running_var of every BatchNorm2d as views of one buffer, num_batches_tracked likewise: the counter of all the
        layers advances with ONE add per training forward."""
        bns = [u.bn for u in self._units if u.bn is not None]
        if not bns:                                          # a module without BatchNorm (CrissCrossAttention)
            self._stat_flat, self._nbt = torch.empty(0, dtype=torch.float32, device=dev), torch.zeros(0, dtype=torch.int64, device=dev)
            self._stat_gen += 1
            return
        n = sum(b.num_features for b in bns)
        flat = torch.empty(2 * n, dtype=torch.float32, device=dev)
        nbt = torch.empty(len(bns), dtype=torch.int64, device=dev)
        off = 0
        with torch.no_grad():
            for i, b in enumerate(bns):
                c = b.num_features
                for name, o in (("running_mean", off), ("running_var", n + off)):
                    v = flat[o:o + c]
                    v.copy_(getattr(b, name))
                    getattr(b, name).data = v          # keeps the buffer object (state_dict / load_state_dict see the view)
                nbt[i] = b.num_batches_tracked
                b.num_batches_tracked.data = nbt[i]
                off += c
        self._stat_flat, self._nbt = flat, nbt
        self._stat_gen += 1

    def _buffers_intact(self, dev):
        u = next((x for x in self._units if x.bn is not None), None)
        if u is None:
            return self._stat_flat is not None and self._stat_flat.device == dev
        return self._stat_flat is not None and self._stat_flat.device == dev and u.bn.running_mean.data_ptr() == self._stat_flat.data_ptr()

    def _build_pack_plan(self, dev):
        rows, off, blk = [], 0, 0
        packed = [u for u in self._units if not u.depthwise]          # depthwise kernels read the fp32 [C,1,3,3] weights directly
        for u in packed:
            kh, kw = u.geom[0], u.geom[1]
            n = gk.pack_elems(u.cout, u.cin, kh, kw)
            rows.append([u.weight._mi_off, off, off, u.cout, u.cin, kh * kw, blk, 0])
            blk += -(-n // 1024)
            off += n
        self._wp_flat = torch.empty(off, dtype=torch.bfloat16, device=dev)
        self._wpt_flat = torch.empty(off, dtype=torch.bfloat16, device=dev)
        for u, r in zip(packed, rows):
            n = gk.pack_elems(u.cout, u.cin, u.geom[0], u.geom[1])
            u.wp = self._wp_flat[r[1]:r[1] + n]
            u.wpt = self._wpt_flat[r[2]:r[2] + n]
        self._pack_blocks = blk
        self._pack_n = len(packed)
        self._pack_table = torch.tensor(rows, dtype=torch.int64, device=dev)

    def _prepare(self):
        st = self.ensure_flat()
        sig = (st.generation, sum(u.weight._version for u in self._units), st.data.data_ptr())
        if sig != self._pack_sig:
            if self._pack_n:                                 # (a module of depthwise convs only has nothing to pack: LocalAttenModule)
                gk.gconv_pack_multi(st.data, self._wp_flat, self._wpt_flat, self._pack_table, self._pack_n, self._pack_blocks)
            self._pack_sig = sig
        return st

    def _grad_slot(self, p):
        """(where d loss / d p is written, whether to accumulate: True from the second write of a backward pass on - shared parameters)."""
        st = self._store
        acc = id(p) in st.written
        st.written.add(id(p))
        g = st.grad[p._mi_off:p._mi_off + p.numel()].view_as(p)
        if p.grad is None or p.grad.data_ptr() != g.data_ptr():      # a zero_grad(set_to_none=True) dropped the view
            p.grad = g
        return g, acc

    def _grad_of(self, p):
        return self._grad_slot(p)[0]

    def zero_grad(self, set_to_none=True):
        """Gradient slots are overwritten by the next backward pass: forget which ones were written instead of clearing 100+ MB."""
        if self._store is not None:
            self._store.written.clear()
        else:
            super().zero_grad(set_to_none)

    def _ones(self, c):
        t = self._ones_cache.get(c)
        if t is None or t.device != self._store.data.device:
            t = self._ones_cache[c] = torch.ones(c, dtype=torch.float32, device=self._store.data.device)
        return t

    def _zeros(self, c):
        t = self._ones_cache.get(-c)
        if t is None or t.device != self._store.data.device:
            t = self._ones_cache[-c] = torch.zeros(c, dtype=torch.float32, device=self._store.data.device)
        return t

    def _eval_fold(self, u):
        bn = u.bn
        sig = (self._store.generation, self._stat_gen, bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version)
        hit = self._eval_cache.get(u.key)
        if hit is None or hit[0] != sig:
            hit = (sig, gk.gbn_fold(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps))
            self._eval_cache[u.key] = hit
        return hit[1]

    precision = "bf16"

    def set_precision(self, precision):
        """'bf16': the training engine's regime in eval() too (bf16 activations and operands, fp32 accumulation).  'fp32': eval() forwards run in
        the reference's precision (csrc/gf32.hip) - what the testers use by default (TEST.PRECISION), so that the masks they threshold are the
        reference's.  train() forwards are bf16 either way."""
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32', got %r" % (precision,))
        self.precision = precision
        return self

    def _graph(self, run, *inputs):
        raise NotImplementedError

    def _run(self, xs, rec, in_needs):
        for x in xs:
            if not x.is_cuda:
                raise _lib.MiError("%s runs on the MI355X only (got a %s tensor); the CPU restatement is oracle/ref_pranet.py, test infrastructure"
                                   % (type(self).__name__, x.device))
        self._prepare()
        run = self.RUN(self, self.training, rec)
        dt = torch.float32 if run.f32 else torch.bfloat16
        ins = [run.var(self._nhwc_input(x, dt, (need and rec) or not self.PAD_IMAGE), need) for x, need in zip(xs, in_needs)]      # NHWC bf16 (fp32 evaluation: fp32)
        outs = self._graph(run, *ins)
        if self.training:
            self._nbt.add_(1)
            self._stat_gen += 1                       # the kernels update the running statistics through raw pointers: no tensor version moves
        return run, ins, outs

    PAD_IMAGE = False          # True on the whole nets whose first op is the stem conv on the image (PraNet, GCPAEncoder)

    @staticmethod
    def _nhwc_input(x, dt, wants_grad):
        """NCHW module input -> NHWC activation.  A three-channel bf16 image that needs no gradient is stored with EIGHT channels (five zero planes): the stem
        conv then reads one 16-byte vector per pixel and tap instead of sixteen 2-byte loads (its packed weights are zero beyond channel 3 anyway), and its weight
        gradient is computed for eight input channels and cut back (see _conv_backward).  GALD's 3 -> 32 stem at 6 x 720 x 1280: 169 -> ~60 us forward, 320 -> ~100
        us weight gradient."""
        nhwc = x.detach().permute(0, 2, 3, 1)
        if dt == torch.bfloat16 and x.shape[1] == 3 and not wants_grad and os.environ.get("MI_STEM_PAD8", "1") != "0":
            out = torch.zeros((x.shape[0], x.shape[2], x.shape[3], 8), dtype=dt, device=x.device)
            out[..., :3] = nhwc
            return out
        return nhwc.to(dt).contiguous()

    def forward(self, *xs):
        self._grad_mode = torch.is_grad_enabled()          # (inside Function.forward grad mode is always off: ask here whether a tape is wanted at all)
        out = _EngineFn.apply(self, len(xs), *xs, *[p for _, p in self.engine_parameters()])
        return out[0] if len(out) == 1 else out


class _EngineFn(torch.autograd.Function):
    """A module of this file as one autograd node: forward records the tape, backward replays it; parameter gradients go straight into the
    flat gradient buffer that every `p.grad` is a view of (autograd receives None for them), input gradients are returned."""

    @staticmethod
    def forward(ctx, net, n_in, *args):
        xs = args[:n_in]
        in_needs = ctx.needs_input_grad[2:2 + n_in]
        rec = any(ctx.needs_input_grad[2:]) and getattr(net, "_grad_mode", True)
        run, ins, outs = net._run(xs, rec, in_needs)
        ctx.run, ctx.ins, ctx.outs, ctx.n_in, ctx.in_dtypes = run, ins, outs, n_in, [x.dtype for x in xs]
        return tuple(o.t.permute(0, 3, 1, 2) if o.t.dim() == 4 else o.t for o in outs)          # NCHW-shaped views of NHWC memory (or scalars: losses)

    @staticmethod
    def backward(ctx, *gouts):
        run = ctx.run
        for o, g in zip(ctx.outs, gouts):
            if g is not None:
                o.g, o.own = (g.permute(0, 2, 3, 1).to(o.t.dtype).contiguous() if g.dim() == 4 else g), True
        run.backward()
        run.net._store.zero_stale()          # parameters this pass did not reach must not keep the previous pass's gradient
        gin = [None if (v.g is None) else v.g.permute(0, 3, 1, 2).to(dt) for v, dt in zip(ctx.ins, ctx.in_dtypes)]
        ctx.run = ctx.ins = ctx.outs = None
        return (None, None) + tuple(gin) + (None,) * (len(ctx.needs_input_grad) - 2 - ctx.n_in)


class Bottle2neck(_Engine):
    """Res2Net_v1b.py:15-92 as a stand-alone module (same constructor arguments and state_dict keys); `downsample`: True builds the
    AvgPool2d(stride, stride, ceil_mode=True, count_include_pad=False) + 1x1 conv + BatchNorm2d path of Res2Net_v1b.py:120-127."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, baseWidth=26, scale=4, stype="normal"):
        super().__init__()
        if scale != 4:
            raise NotImplementedError("the PraNet trunk is 26w x 4s")
        width = int(math.floor(planes * (baseWidth / 64.0)))
        blk = dict(name="", width=width, stride=stride, stage=stype == "stage", down=None)
        blk["conv1"] = _Unit("conv1", "bn1", inplanes, width * scale, 1)
        blk["convs"] = [_Unit("convs.%d" % i, "bns.%d" % i, width, width, 3, stride, 1) for i in range(scale - 1)]
        blk["conv3"] = _Unit("conv3", "bn3", width * scale, planes * 4, 1)
        order = [blk["conv1"]] + blk["convs"] + [blk["conv3"]]
        if downsample:
            blk["down"] = _Unit("downsample.1", "downsample.2", inplanes, planes * 4, 1)
            order.append(blk["down"])
        self._blk = blk
        self._register(order)

    def _graph(self, run, x):
        return [_bottle2neck(run, x, self._blk)]


class RFB_modified(_Engine):
    """PraNet_Res2Net.py:22-59."""

    def __init__(self, in_channel, out_channel):
        super().__init__()
        self._u, flat = _rfb_units("", in_channel, out_channel)
        for u in flat:
            u.key, u.bnkey = u.key.lstrip("."), u.bnkey.lstrip(".")
        self._c = out_channel
        self._register(flat)

    def _graph(self, run, x):
        return [_rfb_block(run, x, self._u, self._c)]


class aggregation(_Engine):
    """PraNet_Res2Net.py:61-96: forward(x1, x2, x3), x1 coarsest; one-channel fp32 output."""

    def __init__(self, channel):
        super().__init__()
        self._a, order = _agg_units("", channel)
        self._c = channel
        self._register(order)

    def _graph(self, run, x1, x2, x3):
        return [_aggregation(run, self._a, self._c, x1, x2, x3)]


class PraNet(_Engine):
    """PraNet(channel=32) of PraNet_Res2Net.py:98-179.  forward(x [B,3,H,W]) -> (lateral_map_5, lateral_map_4, lateral_map_3, lateral_map_2),
    each [B,1,H,W] fp32 logits.  The reference loads ImageNet weights from a local file the image does not have; weights here are
    initialised like the reference's modules (kaiming_normal fan_out for the trunk's convs, Conv2d defaults elsewhere) or loaded from a
    checkpoint / the formula generator."""

    PAD_IMAGE = True

    def __init__(self, channel=32):
        super().__init__()
        c = channel
        self.channel = c
        trunk, self._blocks = _res2net_units()
        self._stem = trunk[:3]
        self._rfb = {}
        order = list(trunk)          # then Res2Net's classifier head: in the reference's state_dict, never run by PraNet
        order += [("resnet.fc.weight", torch.empty(1000, 2048).uniform_(-1, 1) / math.sqrt(2048)), ("resnet.fc.bias", torch.empty(1000).uniform_(-1, 1) / math.sqrt(2048))]
        for name, cin in (("rfb2_1", 512), ("rfb3_1", 1024), ("rfb4_1", 2048)):
            self._rfb[name], flat = _rfb_units(name, cin, c)
            order += flat
        self._agg, agg_order = _agg_units("agg1.", c)
        order += agg_order
        self._ra = {4: [_basic("ra4_conv1", 2048, 256, 1)] + [_basic("ra4_conv%d" % i, 256, 256, 5, 2) for i in (2, 3, 4)] + [_basic("ra4_conv5", 256, 1, 1)],
                    3: [_basic("ra3_conv1", 1024, 64, 1), _basic("ra3_conv2", 64, 64, 3, 1), _basic("ra3_conv3", 64, 64, 3, 1), _basic("ra3_conv4", 64, 1, 3, 1)],
                    2: [_basic("ra2_conv1", 512, 64, 1), _basic("ra2_conv2", 64, 64, 3, 1), _basic("ra2_conv3", 64, 64, 3, 1), _basic("ra2_conv4", 64, 1, 3, 1)]}
        order += self._ra[4] + self._ra[3] + self._ra[2]
        self._register(order)

    def _graph(self, run, x):
        if x.t.shape[1] % 32 or x.t.shape[2] % 32:
            raise _lib.MiError("PraNet input sides must be multiples of 32 (the reverse-attention branches resize by exact factors), got %s" % (tuple(x.t.shape),))
        s = self._stem
        y = run.tap("stem0", run.conv_bn(x, s[0], True))
        y = run.tap("stem1", run.conv_bn(y, s[1], True))
        y = run.tap("stem", run.stem_tail(y, s[2]))
        ends = {}
        for blk in self._blocks:
            y = run.tap(blk["name"], _bottle2neck(run, y, blk))
            ends[blk["name"].rsplit(".", 1)[0]] = y                                  # the last block of each layer wins
        x2, x3, x4 = ends["resnet.layer2"], ends["resnet.layer3"], ends["resnet.layer4"]
        x2r = _rfb_block(run, x2, self._rfb["rfb2_1"], self.channel)
        x3r = _rfb_block(run, x3, self._rfb["rfb3_1"], self.channel)
        x4r = _rfb_block(run, x4, self._rfb["rfb4_1"], self.channel)
        run.tap("rfb2", x2r), run.tap("rfb3", x3r), run.tap("rfb4", x4r)
        coarse = run.tap("coarse", _aggregation(run, self._agg, self.channel, x4r, x3r, x2r))           # ra5_feat: 1/8 resolution, one channel, fp32
        rs = lambda v, f: run.resize(v, f, False)                                    # F.interpolate(..., mode='bilinear'): align_corners False
        maps = [rs(coarse, 8)]
        g = run.tap("ra4", _reverse_branch(run, rs(coarse, 0.25), x4, self._ra[4]))
        maps.append(rs(g, 32))
        g = run.tap("ra3", _reverse_branch(run, rs(g, 2), x3, self._ra[3]))
        maps.append(rs(g, 16))
        g = run.tap("ra2", _reverse_branch(run, rs(g, 2), x2, self._ra[2]))
        maps.append(rs(g, 8))
        return [run.tap("map%d" % i, m) for i, m in enumerate(maps)]


# ------------------------------------------------------------------------------------------------ optimizer / schedule / trainer / tester
class FlatAdam(torch.optim.Adam):
    """torch.optim.Adam(lr) over the module's flat parameter buffer with `clip_gradient(optimizer, clip)` (core/utils/utils.py:6-16) fused in:
    ONE launch per step (mi_adam_step_clamped) instead of one per tensor.  torch's state_dict format (per-parameter exp_avg / exp_avg_sq are
    views of the flat moment buffers).  Parameters the backward pass never writes (Res2Net's unused fc) keep a zero gradient: their moments
    stay zero and they do not move, like the reference's (whose fc.grad is None)."""

    def __init__(self, net, lr, grad_clamp=None):
        self.net = net
        super().__init__(net.parameters(), lr)
        self.grad_clamp = grad_clamp
        self._m = self._v = None
        self._steps = 0
        self.device_hyper = None         # 6-float device tensor (lr, beta1, beta2, eps, clamp, step): HIP-graph mode

    def set_device_hyper(self, enable=True):
        """Graph mode: step() reads its hyper-parameters and the step count from device memory (mi_adam_step_dev); push_hyper() refreshes the
        learning rate from param_groups before a replay."""
        if not enable:
            self.device_hyper = None
            return
        self._ensure_moments()           # (allocated inside a capture they would be re-zeroed by every replay)
        g = self.param_groups[0]
        self.device_hyper = torch.tensor([g["lr"], g["betas"][0], g["betas"][1], g["eps"], self.grad_clamp or 0.0, float(self._steps)], dtype=torch.float32,
                                         device=self.net._store.data.device)

    def push_hyper(self):
        self.device_hyper[0:1].fill_(float(self.param_groups[0]["lr"]))

    def zero_grad(self, set_to_none=True):
        st = self.net._store
        if st is not None:
            st.written.clear()          # every gradient slot is overwritten by the next backward: no 130 MB memset

    def _ensure_moments(self):
        st = self.net.ensure_flat()
        if self._m is None or self._m.numel() != st.data.numel() or self._m.device != st.data.device:
            self._m, self._v = torch.zeros_like(st.data), torch.zeros_like(st.data)
            for p in st.params:
                s = self.state[p]
                if "exp_avg" in s:                                   # restored by load_state_dict: adopt
                    self._m[p._mi_off:p._mi_off + p.numel()].copy_(s["exp_avg"].reshape(-1))
                    self._v[p._mi_off:p._mi_off + p.numel()].copy_(s["exp_avg_sq"].reshape(-1))
                    self._steps = int(s["step"])
                s["exp_avg"] = self._m[p._mi_off:p._mi_off + p.numel()].view_as(p)
                s["exp_avg_sq"] = self._v[p._mi_off:p._mi_off + p.numel()].view_as(p)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        st = self._ensure_moments()
        g = self.param_groups[0]
        if g.get("amsgrad") or g.get("weight_decay", 0) != 0 or g.get("maximize"):
            raise NotImplementedError("FlatAdam implements the reference's configuration (pranet_trainer.py:20)")
        self._steps += 1
        # a backward pass that never reached this module (detached features, a loss that bypasses it) ran no zero_stale(): the previous pass's gradients
        # would still sit in the flat buffer and be applied.  Cleared here: what torch's zero_grad(set_to_none=False) leaves (a no-op after a normal pass)
        st.zero_stale()
        if self.device_hyper is not None:
            if not torch.cuda.is_current_stream_capturing():
                self.push_hyper()
            self.device_hyper[5:6].add_(1.0)                           # (captured with the step: every replay advances the device-side count)
            K.adam_step_dev(st.data, st.grad, self._m, self._v, self.device_hyper)
        else:
            K.adam_step(st.data, st.grad, self._m, self._v, g["lr"], g["betas"][0], g["betas"][1], g["eps"], self._steps, grad_clamp=self.grad_clamp)
        st.generation += 1

    def state_dict(self):
        step_t = torch.tensor(float(self._steps))                      # (graph replays advance the count without running step())
        for p, s in self.state.items():
            if "exp_avg" in s:
                s["step"] = step_t
        return super().state_dict()


class GraphedStep:
    """One whole optimizer step of pranet_trainer.py:39-60 - weight pack, forward, four structure losses, backward, clamped Adam: ~1 800
    launches - captured once as a HIP graph and replayed: the host cost of a step becomes one graph launch.  Inputs are copied into static
    buffers; the learning rate and Adam's step count live in device memory (FlatAdam.set_device_hyper)."""

    def __init__(self, net, opt, images, gts, warmup=3):
        self.net, self.opt = net, opt
        self.x, self.gt = images.detach().clone(), gts.detach().clone()
        net.ensure_flat()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                  # eager warm-up on the capture stream's pool: workspaces, moment buffers, caches
            for _ in range(warmup):
                self._core()
        torch.cuda.current_stream().wait_stream(side)
        opt.set_device_hyper(True)
        torch.cuda.synchronize()
        net._pack_sig = None                           # the weight pack belongs inside the graph: weights change on every replay
        steps = opt._steps
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.losses = self._core()
        opt._steps = steps                             # the capture itself executed nothing

    def _core(self):
        self.opt.zero_grad()
        ls = [structure_loss(o, self.gt) for o in self.net(self.x)]
        (ls[3] + ls[2] + ls[1] + ls[0]).backward()
        self.opt.step()
        return [l.detach() for l in ls]

    def __call__(self, images=None, gts=None):
        """Returns the four losses (lateral 5, 4, 3, 2) as device scalars owned by the graph (clone to keep across replays)."""
        if images is not None:
            self.x.copy_(images, non_blocking=True)
            self.gt.copy_(gts, non_blocking=True)
        self.opt.push_hyper()
        self.graph.replay()
        self.opt._steps += 1
        self.net._stat_gen += 1
        self.net._store.generation += 1
        return self.losses


def graph_mode_default():
    """PraNetTrainer replays its optimizer step as a HIP graph unless MI_GRAPH=0: ~1 300 launches of 3 - 40 us per step leave the host 15 - 20 % behind the
    GPU when enqueued one by one (bench.py prints both numbers); the replay is bit-equal to eager (tests/test_gpu_pranet.py)."""
    return os.environ.get("MI_GRAPH", "1") != "0"


def warmup_cosine_lr(base_lr, steps, multiplier=8.0, warm=5, t_max=100):
    """Learning rate after `steps` calls of scheduler.step() in pranet_trainer.py:97-104: GradualWarmupScheduler(multiplier=8, total_epoch=5)
    (core/utils/adapt_lr.py:19-45) climbs linearly from base_lr to 8 * base_lr over 5 epochs, then hands over to CosineAnnealingLR(T_max=100,
    eta_min=0).  The hand-over as the reference's chain performs it (pinned by tests/golden/g12_pranet_lr.npz, written by running that chain):
    the cosine scheduler continues RECURSIVELY from the group's current rate 8 * base_lr with its own epoch counter already at 1, so epoch
    warm + 1 overshoots to 8 * base_lr * 2 / (1 + cos(pi / t_max)) and every later rate keeps that factor over the textbook closed form."""
    if steps <= warm:
        return base_lr * ((multiplier - 1.0) * steps / warm + 1.0)
    t = steps - warm - 1
    return base_lr * multiplier * (1.0 + math.cos(math.pi * t / t_max)) / (1.0 + math.cos(math.pi / t_max))


def clip_gradient(optimizer, grad_clip):
    """core/utils/utils.py:6-16 (API parity; PraNetTrainer fuses the clamp into FlatAdam's update instead)."""
    for group in optimizer.param_groups:
        for param in group["params"]:
            if param.grad is not None:
                param.grad.data.clamp_(-grad_clip, grad_clip)


class AvgMeter:
    """core/utils/utils.py:18-38."""

    def __init__(self, num=40):
        self.num = num
        self.losses = []

    def update(self, val, n=1):
        self.losses.append(val)

    def show(self):
        return torch.mean(torch.stack(self.losses[max(len(self.losses) - self.num, 0):]))


class PraNetTrainer(BaseTrainer):
    """pranet_trainer.py:12-104: Adam(BASE_LR / 8), three passes per batch (the reference's multi-scale loop, whose rescale is a no-op:
    it resizes to trainsize whatever the rate - Appendix B), structure loss on the four side outputs, gradient clamp 0.5, warm-up + cosine
    schedule per epoch, checkpoint {'epoch', 'model', 'optimizer'} as PraNet-<epoch>.pth."""

    def __init__(self, name, cfg, train_loader, local_rank, logger=None):
        super().__init__(name, cfg, train_loader, local_rank, logger)

    def init_params(self):
        self.trainsize = self.cfg.INPUT.TRAINSIZE
        self.model = PraNet().to(self.device)
        self.model.ensure_flat()
        self.base_lr = self.cfg.SOLVER.BASE_LR / 8
        self.optimizer = FlatAdam(self.model, self.base_lr, grad_clamp=0.5)

    structure_loss = staticmethod(structure_loss)

    def _resize_to_trainsize(self, t):
        """F.upsample(t, size=(trainsize, trainsize), mode='bilinear', align_corners=True) of pranet_trainer.py:47-48."""
        if t.shape[-2:] == (self.trainsize, self.trainsize):
            return t                                               # same size with align_corners=True: the identity
        nhwc = t.float().permute(0, 2, 3, 1).contiguous()
        return gk.gresize(nhwc, (self.trainsize, self.trainsize), True).permute(0, 3, 1, 2).contiguous()

    GRAPH_WARMUP = 3

    def train_step(self, images, gts):
        """One optimizer step (pranet_trainer.py:39-60).  Returns the four losses (lateral 5, 4, 3, 2) as device scalars.
        After three eager steps the step is captured as a HIP graph and replayed while the input shape stays the same (MI_GRAPH=0: always eager)."""
        if graph_mode_default():
            st = self.__dict__.setdefault("_graph", {"eager": 0, "step": None, "shape": None})
            if st["step"] is not None and st["shape"] == (tuple(images.shape), tuple(gts.shape)):
                return [l.clone() for l in st["step"](images, gts)]
            if st["step"] is None and st["eager"] >= self.GRAPH_WARMUP:
                st["step"], st["shape"] = GraphedStep(self.model, self.optimizer, images, gts, warmup=0), (tuple(images.shape), tuple(gts.shape))
                return [l.clone() for l in st["step"](images, gts)]
            st["eager"] += 1
        self.optimizer.zero_grad()
        outs = self.model(images)
        losses = [self.structure_loss(o, gts) for o in outs]
        loss = losses[3] + losses[2] + losses[1] + losses[0]
        loss.backward()
        self.optimizer.step()                                       # clip_gradient(optimizer, 0.5) is fused into the update
        return losses

    def _train_epoch(self, epoch):
        size_rates = [0.75, 1, 1.25]
        rec = [AvgMeter() for _ in range(4)]                        # lateral 2, 3, 4, 5
        n = len(self.train_loader)
        for i, pack in enumerate(self.train_loader):
            for rate in size_rates:
                images, gts, _ = pack
                images = images.to(self.device, non_blocking=True)
                gts = gts.to(self.device, non_blocking=True).float()
                if gts.dim() == 3:
                    gts = gts.unsqueeze(1)
                if rate != 1:
                    images, gts = self._resize_to_trainsize(images), self._resize_to_trainsize(gts)
                l5, l4, l3, l2 = self.train_step(images, gts)
                if rate == 1:
                    for m, v in zip(rec, (l2, l3, l4, l5)):
                        m.update(v.detach(), self.cfg.SOLVER.BATCH_SIZE)
            if i % 20 == 0 or i == n:
                self.logger.info("{} Epoch [{:03d}/{:03d}], Step [{:04d}/{:04d}], [lateral-2: {:.4f}, lateral-3: {:0.4f}, lateral-4: {:0.4f}, lateral-5: {:0.4f}, "
                                 "learning_rate: {:0.8f}]".format(datetime.now(), epoch, self.cfg.SOLVER.EPOCHS, i, n, rec[0].show(), rec[1].show(), rec[2].show(),
                                                                  rec[3].show(), self.optimizer.param_groups[0]["lr"]))
        save_path = self.cfg.OUTPUT_DIR
        os.makedirs(save_path, exist_ok=True)
        if epoch % self.cfg.SOLVER.CHECKPOINT_PERIOD == 0:
            self._save_checkpoint(epoch, save_path + "PraNet-%d.pth" % epoch)
            self.logger.info("[Saving Snapshot:] " + save_path + "PraNet-{}.pth".format(epoch))

    def _val_epoch(self, epoch):
        raise NotImplementedError("the reference's PraNetTrainer has no validation epoch")

    def _save_checkpoint(self, epoch, save_path):
        torch.save({"epoch": epoch, "model": self.model.state_dict(), "optimizer": self.optimizer.state_dict()}, save_path)

    def _load_checkpoint(self):
        self.checkpoint = torch.load(self.cfg.resume, map_location=self.device)
        self.model.load_state_dict(self.checkpoint["model"])
        if "optimizer" in self.checkpoint:
            self.logger.info("Loading optimizer from {}".format(self.cfg.resume))
            self.optimizer.load_state_dict(self.checkpoint["optimizer"])
            self.optimizer._m = None                                  # re-adopt the restored moments on the next step
            self.optimizer._ensure_moments()
        if "epoch" in self.checkpoint:
            self.start_epoch = self.checkpoint["epoch"] + 1

    def train(self):
        self.model.train()
        self.logger.info("#" * 20 + " Start Training " + "#" * 20)
        for k, epoch in enumerate(range(self.start_epoch, self.cfg.SOLVER.EPOCHS + 1)):
            for grp in self.optimizer.param_groups:                   # the reference builds fresh schedulers per run: k steps since start
                grp["lr"] = warmup_cosine_lr(self.base_lr, k)
            self._train_epoch(epoch)


class PranetTester:
    """pranet_tester.py:10-53: res2 -> resize to the label size (align_corners False) -> sigmoid -> min-max normalise over the batch ->
    {background, polyp} by which of (1 - p, p) is larger -> intersection / union meters."""

    def __init__(self, cfg, device, test_loader, logger):
        self.cfg, self.logger, self.test_loader, self.device = cfg, logger, test_loader, device
        self.model = PraNet()
        self.model.to(device)
        # The reference thresholds an fp32 forward (pranet_tester.py:36-46): TEST.PRECISION 'fp32' (default) evaluates in the reference's precision
        # (csrc/gf32.hip: maps within 1e-5 of the reference's, masks identical), 'bf16' in the training engine's regime (faster).
        self.model.set_precision(cfg.TEST.PRECISION if "PRECISION" in cfg.TEST else "fp32")

    def _load_checkpoint(self):
        self.logger.info("Loading checkpoint from {}".format(self.cfg.resume))
        checkpoint = torch.load(self.cfg.resume, map_location=self.device)
        self.model.load_state_dict(checkpoint["model"])

    def predict(self, x, hw):
        with torch.no_grad():
            res2 = self.model(x)[3].float()
            out = gk.gresize(res2.permute(0, 2, 3, 1).contiguous(), hw, False).permute(0, 3, 1, 2)
            p = out.sigmoid().squeeze(1)
            p = (p - p.min()) / (p.max() - p.min() + 1e-8)
            return (p > 1 - p).long()                                  # np.stack([1 - p, p]).max(1)[1]: ties go to background

    def test(self):
        from .metrics import AverageMeter, intersectionAndUnionGPU
        self.model.eval()
        self.meter = AverageMeter()
        for x, y, _ in self.test_loader:
            x = x.to(self.device, non_blocking=True)
            y = y.to(self.device, non_blocking=True)
            h, w = y.shape[-2:]
            y = y.reshape(y.shape[0], h, w).long()
            pred = self.predict(x, (h, w))
            inter, union, target, res = intersectionAndUnionGPU(pred, y, self.cfg.MODEL.NUM_CLASSES, self.cfg.INPUT.IGNORE_LABEL)
            self.meter.update(inter.cpu().numpy(), union.cpu().numpy(), target.cpu().numpy(), res.cpu().numpy())
        self.meter.summary(self.logger, self.cfg.MODEL.NUM_CLASSES)
