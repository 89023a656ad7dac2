"""The MI355X execution engine of the DeepLabV2 hot path: a static schedule of C-ABI kernel launches
(forward AND hand-written backward) over NHWC bf16 buffers, exposed to PyTorch as autograd Functions.

What it replaces in the reference: the eager op-by-op execution of
core/components/resnet.py:93-113 (Bottleneck.forward: conv, FrozenBN, ReLU, residual) for every
block of layer1..layer4, core/models/classifiers/aspp/classifier.py:26-32 (ASPP forward) and the
autograd-generated backward of both, plus `criterion(output, label)` of
core/trainers/aspp_trainer.py:89-92.

Design: one process per GPU owns the schedule.  FrozenBN (+ReLU, +residual) lives in the GEMM
epilogues; in backward the ReLU masks are applied in the epilogue of the data-gradient GEMM that
produces the tensor, and FrozenBN scales are folded into the packed dgrad weights / the wgrad
reduction.  Weight gradients are written by the wgrad reduce kernel straight into each
parameter's `.grad` (a view of one flat fp32 buffer per module -> fused SGD and bucketed RCCL
all-reduce work on contiguous ranges).  No tensor of the schedule ever visits the CPU.
"""
import contextlib
import os

import torch

from .. import kernels as K
from . import arch


# ------------------------------------------------------------------------------------------------ flat storage
class FlatStore:
    """One contiguous fp32 buffer for a module's parameters and one for their gradients."""

    ALIGN = 64  # floats (256 B): every parameter starts on a cache-line boundary

    def __init__(self, named_params, device):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += -(-p.numel() // self.ALIGN) * self.ALIGN
        self.total = off
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.written = set()      # ids of params whose .grad the engine has overwritten since zero_grad
        self.dirty = set()        # ids of params whose slot holds a gradient of SOME backward pass (tape engines: see zero_stale)
        self.generation = 0       # bumped by FusedSGD.step (raw-pointer updates do not bump torch versions)
        self.grad_hooks = []      # callables(store, lo, hi) fired when grads [lo, hi) are final (DDP)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                v = self.data[o:o + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
                p.grad = self.grad[o:o + p.numel()].view_as(p)
                p._mi_store = self
                p._mi_off = o

    def zero_stale(self):
        """zero_grad() of the tape engines only forgets which slots were written (the next backward overwrites them).  A backward pass that
        does NOT reach a parameter (a loss on some outputs only, an input branch without gradient) would leave the previous pass's values in its
        slot for the optimizer to apply: after a pass, slots written by an earlier pass and not by this one are cleared - what torch's
        zero_grad(set_to_none=False) leaves there."""
        stale = self.dirty - self.written
        if stale:
            for p in self.params:
                if id(p) in stale:
                    self.grad[p._mi_off:p._mi_off + p.numel()].zero_()
        self.dirty = set(self.written)

    def owns(self, p):
        return getattr(p, "_mi_store", None) is self and p.data_ptr() == self.data.data_ptr() + 4 * p._mi_off

    def intact(self):
        return all(self.owns(p) for p in self.params)

    def grad_view(self, p):
        return self.grad[p._mi_off:p._mi_off + p.numel()].view_as(p)

    def span(self, params):
        """[lo, hi) float range covering `params` (must be consecutive in this store)."""
        lo = min(p._mi_off for p in params)
        hi = max(p._mi_off + -(-p.numel() // self.ALIGN) * self.ALIGN for p in params)
        return lo, hi


def grad_slot(p):
    """Where the engine writes d loss / d p, and whether to accumulate into it."""
    st = getattr(p, "_mi_store", None)
    if st is not None and st.owns(p):
        want = st.grad.data_ptr() + 4 * p._mi_off
        if p.grad is None or p.grad.data_ptr() != want:
            p.grad = st.grad_view(p)          # a foreign optimizer's zero_grad(set_to_none=True) dropped it
            st.written.discard(id(p))
        acc = id(p) in st.written
        st.written.add(id(p))
        return p.grad, acc
    if p.grad is None:
        p.grad = torch.empty_like(p, memory_format=torch.contiguous_format)
        return p.grad, False
    return p.grad, True


# ------------------------------------------------------------------------------------------------ weight-gradient stream
class _SideStream:
    """Weight gradients are off the backward critical path (nothing consumes them before the optimizer / all-reduce), so they
    run on a second HIP stream: their workgroups fill the CUs that the data-gradient kernels leave idle in their last partial
    round (tile quantisation costs 10-25 % of a launch at M = 75 272), and the small slab reducers hide entirely.
    MI_WGRAD_STREAM=0 puts everything back on one stream."""
    _by_device = {}

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)          # (high priority for it measured 288.4 vs 290.7 images/s: profiles/r05_wgrad_sched_ab.txt)
        self.dirty = False

    @classmethod
    def get(cls, device):
        if os.environ.get("MI_WGRAD_STREAM", "1") == "0" or device.type != "cuda":
            return None
        if K.PROFILE is not None:      # per-launch timing requested (bench.py's instrumented steps): one stream, so that a
            return None                # launch's event-to-event time is that kernel's alone
        key = device.index if device.index is not None else torch.cuda.current_device()
        if key not in cls._by_device:
            cls._by_device[key] = cls(torch.device("cuda", key))
        return cls._by_device[key]

    def run(self, fn, *inputs):
        """fn() on the side stream after everything enqueued so far on the current stream; `inputs` are the tensors it reads
        (their memory must not be recycled by the allocator before the side stream is done with them)."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            fn()
        for t in inputs:
            t.record_stream(self.stream)
        self.dirty = True

    def join(self):
        if self.dirty:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.dirty = False


def _off_path(side, fn, *inputs):
    if side is None:
        fn()
    else:
        side.run(fn, *inputs)


class _Lanes:
    """The two halves of the batch run as two kernel chains on two HIP streams (samples are independent in a FrozenBN net).
    A conv launch at M = 75 272 leaves 8-25 % of its last round of workgroups empty and runs its prologue / epilogue phases
    chip-wide in step; with a second, independent chain in flight the other half's workgroups fill those holes: the backbone
    forward takes 9.6 ms instead of 10.8 ms (`tools/twostream_fwd.py`; four lanes are slower again), +3 % images/s on the
    training step.  Forward only: in backward the weight-gradient stream already supplies the second chain, and splitting the
    data-gradient chain as well measured 4 % slower (three streams' workgroups evict each other's L2 lines).
    MI_BATCH_LANES=1 disables it."""
    _by_device = {}

    @classmethod
    def get(cls, device, batch):
        if os.environ.get("MI_BATCH_LANES", "2") != "2" or device.type != "cuda" or batch < 2 or batch % 2 or K.PROFILE is not None:
            return None
        key = device.index if device.index is not None else torch.cuda.current_device()
        if key not in cls._by_device:
            cls._by_device[key] = torch.cuda.Stream(device=torch.device("cuda", key))
        return cls._by_device[key]


def bn_fold(bn):
    """(scale, shift) of a normalisation layer evaluated on its stored statistics: FrozenBatchNorm2d (layers.py:18-20, no eps) or an
    nn.BatchNorm2d in eval mode (scale = gamma * rsqrt(running_var + eps))."""
    eps = float(getattr(bn, "eps", 0.0) or 0.0)
    var = bn.running_var if eps == 0.0 else bn.running_var + eps
    return K.frozen_bn_fold(bn.weight.detach(), bn.bias.detach(), bn.running_mean, var)


def bn_versions(bn):
    return bn.weight._version + bn.bias._version + bn.running_mean._version + bn.running_var._version


# ------------------------------------------------------------------------------------------------ backbone stages
class _ConvRT:
    __slots__ = ("spec", "weight", "bn", "scale", "shift", "wp", "wpt")

    def __init__(self, spec, weight, bn):
        self.spec, self.weight, self.bn = spec, weight, bn
        self.scale = self.shift = self.wp = self.wpt = None


class StageEngine:
    """layer1..layer4 of the dilated ResNet as a kernel schedule."""

    def __init__(self, owner, plan):
        self.owner = owner                     # the feature-extractor module (holds params/buffers)
        self.plan = plan
        self.blocks = []
        for blk in plan:
            rts = []
            for c in arch.block_convs(blk):
                rts.append(_ConvRT(c, arch.node_at(owner.backbone, c.key).weight, arch.node_at(owner.backbone, c.bn)))
            self.blocks.append((blk, rts))
        self.convs = [rt for _, rts in self.blocks for rt in rts]
        self._bn_sig = None
        self._pack_sig = None
        self._have_dgrad = False
        self.batch_stats = False               # True: trainable BatchNorm2d in train(): no fold, unscaled data-gradient operands

    # -- preparation: FrozenBN folds and bf16 operand packs, redone only when their sources changed
    def _signature(self):
        store = getattr(self.convs[0].weight, "_mi_store", None)
        gen = store.generation if store is not None else -1
        return (gen, sum(rt.weight._version for rt in self.convs), self.convs[0].weight.data_ptr())

    def _build_pack_plan(self, store):
        """Flat bf16 buffers for all packed operands + the device table of mi_pack_weights_multi."""
        dev = store.data.device
        n = sum(rt.weight.numel() for rt in self.convs)
        self._wp_flat = torch.empty(n, dtype=torch.bfloat16, device=dev)
        self._wpt_flat = torch.empty(n, dtype=torch.bfloat16, device=dev)
        self._scale_flat = torch.empty(sum(rt.spec.cout for rt in self.convs), dtype=torch.float32, device=dev)
        self._shift_flat = torch.empty_like(self._scale_flat)
        rows, off, soff, blk = [], 0, 0, 0
        for rt in self.convs:
            c = rt.spec
            T = c.k * c.k
            assert T <= 9, "mi_pack_weights_multi packs 1x1 and 3x3 convs"
            rows.append([rt.weight._mi_off, soff, off, off, c.cout, c.cin, T, blk])
            rt.wp = self._wp_flat[off:off + rt.weight.numel()].view(T, c.cout, c.cin)
            rt.wpt = self._wpt_flat[off:off + rt.weight.numel()].view(T, c.cin, c.cout)
            rt.scale = self._scale_flat[soff:soff + c.cout]
            rt.shift = self._shift_flat[soff:soff + c.cout]
            off += rt.weight.numel()
            soff += c.cout
            blk += -(-c.cout // 32) * -(-c.cin // 128)        # one block per 32 (o) x 128 (i) channels
        self._pack_blocks = blk
        self._table_train = torch.tensor(rows, dtype=torch.int64, device=dev)
        self._table_batch = torch.tensor([r[:1] + [-1] + r[2:] for r in rows], dtype=torch.int64, device=dev)     # no folded scale
        for r in rows:
            r[3] = -1
        self._table_eval = torch.tensor(rows, dtype=torch.int64, device=dev)
        self._plan_store = store

    def prepare(self, train):
        store = getattr(self.convs[0].weight, "_mi_store", None)
        if store is None or not store.intact():
            raise RuntimeError("backbone parameters are not in their flat store; call ensure_flat() after moving the module")
        if getattr(self, "_plan_store", None) is not store:
            self._build_pack_plan(store)
            self._bn_sig = self._pack_sig = None
        refold = False
        if not self.batch_stats:
            bn_sig = (sum(bn_versions(rt.bn) for rt in self.convs), self.convs[0].bn.weight.data_ptr())
            refold = bn_sig != self._bn_sig
            if refold:
                for rt in self.convs:
                    sc, sh = bn_fold(rt.bn)
                    rt.scale.copy_(sc)
                    rt.shift.copy_(sh)
                self._bn_sig = bn_sig
        sig = self._signature() + (self.batch_stats,)
        if sig != self._pack_sig or refold or (train and not self._have_dgrad):
            table = self._table_batch if self.batch_stats else (self._table_train if train else self._table_eval)
            K.pack_weights_multi(store.data, self._scale_flat, self._wp_flat, self._wpt_flat, table, len(self.convs), self._pack_blocks)
            self._pack_sig = sig
            self._have_dgrad = train

    # -- chained 1x1 pairs (csrc/chain.hip): conv3 of a block + conv1 of the next in one launch, and the mirrored pair of data gradients
    @staticmethod
    def chain_mode():
        """MI_CHAIN: 'all' forward and backward, 'fwd', 'bwd', '0' (default) off.  Off by default on numbers: the
        launch alone beats the pair it replaces (151 vs 175 us) but owns its CU (154 KiB of LDS), so the weight-gradient stream and the second forward
        lane stop overlapping with it - the step measured 29.1 ms with it against 28.0 ms without."""
        return os.environ.get("MI_CHAIN", "0")

    def _chain_pair(self, bi):
        """(conv3 of block bi, conv1 of block bi+1) when the pair has the shape mi_conv_chain is built for (256 -> 1024 -> 256, stride 1: the 22
        identity hand-overs of layer3), else None."""
        if bi < 0 or bi + 1 >= len(self.blocks) or self.batch_stats:
            return None
        (_, rts), (nblk, nrts) = self.blocks[bi], self.blocks[bi + 1]
        c3, c1 = rts[2].spec, nrts[0].spec
        ok = (not nblk.down and c3.k == 1 and c1.k == 1 and c3.stride == 1 and c1.stride == 1 and (c3.cin, c3.cout) == (256, 1024)
              and (c1.cin, c1.cout) == (1024, 256))
        return (rts[2], nrts[0]) if ok else None

    @staticmethod
    def _chain_fits(x):
        return x.is_cuda and x.shape[0] * x.shape[1] * x.shape[2] * 2048 < 2 ** 31

    # -- forward
    @staticmethod
    def _fwd_conv(x, rt, relu, res=None, want_mask=False):
        """conv + FrozenBN (+ residual) (+ ReLU).  want_mask: also return the packed sign bits of the output (1 bit per
        element), which is all the backward pass needs of a ReLU'd activation when it is used only as a mask."""
        c = rt.spec
        hw = arch.out_hw(x.shape[1], x.shape[2], c)
        bits = torch.empty((x.shape[0], hw[0], hw[1], c.cout // 16), dtype=torch.int16, device=x.device) if want_mask else None
        y = K.conv_gemm(x, rt.wp, hw, c.k, c.stride, c.pad, c.dil, K.GATHER_FWD, scale=rt.scale, bias=rt.shift, res=res, relu=relu,
                        mask_out=bits)
        return (y, bits) if want_mask else y

    @staticmethod
    def _conv_into(x, rt, relu, y, bits=None, res=None):
        c = rt.spec
        K.conv_gemm(x, rt.wp, (y.shape[1], y.shape[2]), c.k, c.stride, c.pad, c.dil, K.GATHER_FWD, scale=rt.scale, bias=rt.shift, res=res,
                    relu=relu, mask_out=bits, out=y)

    def _forward_lanes(self, x, save, extra):
        """forward() with the two halves of the batch on two streams.  Every output is allocated up front on the caller's stream
        (and is alive until after the join), the lanes only write into their halves."""
        B, dev = x.shape[0], x.device
        halves = (slice(0, B // 2), slice(B // 2, B))
        main = torch.cuda.current_stream()

        def new(like_hw, ch, dtype=torch.bfloat16):
            return torch.empty((B, like_hw[0], like_hw[1], ch), dtype=dtype, device=dev)

        plan, hw = [], (x.shape[1], x.shape[2])
        for blk, rts in self.blocks:
            hw1 = arch.out_hw(hw[0], hw[1], rts[0].spec)
            hw2 = arch.out_hw(hw1[0], hw1[1], rts[1].spec)
            hw3 = arch.out_hw(hw2[0], hw2[1], rts[2].spec)
            t = {"a1": new(hw1, rts[0].spec.cout), "a2": new(hw2, rts[1].spec.cout), "out": new(hw3, rts[2].spec.cout),
                 "idn": new(arch.out_hw(hw[0], hw[1], rts[3].spec), rts[3].spec.cout) if blk.down else None}
            if save:
                t.update(b1=new(hw1, rts[0].spec.cout // 16, torch.int16), b2=new(hw2, rts[1].spec.cout // 16, torch.int16),
                         ob=new(hw3, rts[2].spec.cout // 16, torch.int16))
            plan.append(t)
            hw = hw3
        fork = torch.cuda.Event()
        fork.record(main)
        extra.wait_event(fork)
        xin = x
        chain = save and self.chain_mode() in ("all", "fwd") and self._chain_fits(x)
        ahead = False                              # this block's a1 / b1 were written by the previous block's chained launch
        for bi, ((blk, rts), t) in enumerate(zip(self.blocks, plan)):
            pair = self._chain_pair(bi) if chain else None
            tn = plan[bi + 1] if pair is not None else None
            for lane, h in enumerate(halves):
                with torch.cuda.stream(extra) if lane else contextlib.nullcontext():
                    sl = lambda name: None if t.get(name) is None else t[name][h]
                    if not ahead:
                        self._conv_into(xin[h], rts[0], True, sl("a1"), sl("b1"))
                    self._conv_into(sl("a1"), rts[1], True, sl("a2"), sl("b2"))
                    if blk.down:
                        self._conv_into(xin[h], rts[3], False, sl("idn"))
                    res = sl("idn") if blk.down else xin[h]
                    if pair is not None:
                        K.conv_chain(sl("a2"), pair[0].wp, res, pair[1].wp, pair[0].scale, pair[0].shift, pair[1].scale, pair[1].shift,
                                     mid=sl("out"), out=tn["a1"][h], bits1_out=sl("ob"), bits2_out=tn["b1"][h])
                    else:
                        self._conv_into(sl("a2"), rts[2], True, sl("out"), sl("ob"), res=res)
            ahead = pair is not None
            xin = t["out"]
        main.wait_stream(extra)
        saved, xbits, xin = [], None, x
        if save:
            for t in plan:
                saved.append((xin, t["a1"], t["a2"], xbits, t["b1"], t["b2"]))
                xbits, xin = t["ob"], t["out"]
        return plan[-1]["out"], saved, (plan[-1]["ob"] if save else None)

    def forward(self, x, save):
        """Returns (feature, saved, feature sign bits).  saved[i] = (x, a1, a2, bits(x) or None, bits(a1), bits(a2))."""
        extra = _Lanes.get(x.device, x.shape[0])
        if extra is not None:
            return self._forward_lanes(x, save, extra)
        saved = []
        xbits = None
        chain = save and self.chain_mode() in ("all", "fwd") and self._chain_fits(x)
        ahead = None                               # (a1, bits(a1)) of this block when the previous block's chained launch produced them
        for bi, (blk, rts) in enumerate(self.blocks):
            if ahead is not None:
                a1, b1 = ahead
                a2, b2 = self._fwd_conv(a1, rts[1], True, want_mask=True)
            elif save:
                a1, b1 = self._fwd_conv(x, rts[0], True, want_mask=True)
                a2, b2 = self._fwd_conv(a1, rts[1], True, want_mask=True)
            else:
                a1 = self._fwd_conv(x, rts[0], True)
                a2 = self._fwd_conv(a1, rts[1], True)
                b1 = b2 = None
            idn = self._fwd_conv(x, rts[3], False) if blk.down else x
            pair = self._chain_pair(bi) if chain else None
            ahead = None
            if pair is not None:                   # conv3 + FrozenBN + residual + ReLU, then the next block's conv1 + FrozenBN + ReLU: one launch
                out, a1n, obits, b1n = K.conv_chain(a2, pair[0].wp, idn, pair[1].wp, pair[0].scale, pair[0].shift, pair[1].scale, pair[1].shift)
                ahead = (a1n, b1n)
                saved.append((x, a1, a2, xbits, b1, b2))
                xbits = obits
            elif save:
                out, obits = self._fwd_conv(a2, rts[2], True, res=idn, want_mask=True)
                saved.append((x, a1, a2, xbits, b1, b2))
                xbits = obits
            else:
                out = self._fwd_conv(a2, rts[2], True, res=idn)
            x = out
        return x, saved, xbits

    # -- backward
    @staticmethod
    def _wgrad(dy, xin, rt, batch=None):
        c = rt.spec
        dw, acc = grad_slot(rt.weight)
        K.conv_wgrad(dy, xin, dw, c.k, c.stride, c.pad, c.dil, scale=rt.scale, accumulate=acc, batch=batch)

    @staticmethod
    def _dgrad(dy, rt, in_hw, res=None, bits=None):
        c = rt.spec
        return K.conv_gemm(dy, rt.wpt, in_hw, c.k, c.stride, c.pad, c.dil, K.GATHER_DGRAD, res=res, bits=bits)

    def backward(self, saved, fbits, dfeat, need_dx):
        """dfeat: d loss / d feat (bf16 NHWC); fbits: sign bits of feat.  Returns d loss / d x of the first block."""
        g = K.relu_mask(dfeat, fbits)                     # through the last block's ReLU
        store = getattr(self.convs[0].weight, "_mi_store", None)
        side = _SideStream.get(dfeat.device)
        # MI_BWD_PAIR=1 (experiment, section 8 of DESIGN.md): pair the launches of the two streams by the resource that bounds them.  In stream order a
        # block is D3 D2 D1 on the main stream (data gradients: MFMA-, MFMA-, HBM-bound) and W3 W2 W1 behind them on the side stream (weight gradients:
        # HBM-, MFMA-, HBM-bound), i.e. W2 beside D2 (both MFMA-bound) and W1 beside D1 (both HBM-bound).  Deferring W1 to the next block's start puts
        # W3 + W1' beside D3 D2 and W2 beside D1.
        bwd_pair = side is not None and os.environ.get("MI_BWD_PAIR", "0") == "1"
        deferred = None
        chain = self.chain_mode() in ("all", "bwd") and self._chain_fits(dfeat)
        ga2_ahead = None

        def hooks(rts_):
            if store is not None and store.grad_hooks:
                lo, hi = store.span([rt.weight for rt in rts_])
                with torch.cuda.stream(side.stream) if side is not None else contextlib.nullcontext():
                    for hook in store.grad_hooks:         # the all-reduce of this range is ordered after its weight gradients
                        hook(store, lo, hi)
        # MI_WGRAD_REDUCE_BATCH=1 (default): the slab reducers of a block's weight gradients run as ONE launch at the end of the block (K.WgradBatch)
        # instead of one 10-us launch behind each weight gradient - on the side stream each of those waited ~35 us for its turn beside the data-gradient chain
        batching = os.environ.get("MI_WGRAD_REDUCE_BATCH", "1") != "0" and not bwd_pair and K.PROFILE is None
        # (Tried, profiles/r05_wgrad_sched_ab.txt: the block's three weight gradients on two / three streams - 281 / 280 images/s against 297 on one, the
        #  streams' workgroups evict each other's L2 lines; reducers batched over two blocks - 298.4 against 297.5, inside the noise.  Neither is kept.)
        for bi in range(len(self.blocks) - 1, -1, -1):
            blk, rts = self.blocks[bi]
            x, a1, a2, xb, b1, b2 = saved[bi]
            first = bi == 0
            hw_in = (x.shape[1], x.shape[2])
            hw_mid = (a1.shape[1], a1.shape[2])
            wb = K.WgradBatch() if batching else None
            _off_path(side, lambda: self._wgrad(g, a2, rts[2], wb), g, a2)
            if deferred is not None:                      # the previous block's W1 (and its hooks) ride beside this block's D3 / D2
                dga1, dx_, drts = deferred
                _off_path(side, lambda: self._wgrad(dga1, dx_, drts[0]), dga1, dx_)
                hooks(drts)
                deferred = None
            if ga2_ahead is not None:                     # written by the chained launch of the block above (its conv1's data gradient)
                ga2, ga2_ahead = ga2_ahead, None
            else:
                ga2 = self._dgrad(g, rts[2], (a2.shape[1], a2.shape[2]), bits=b2)
            _off_path(side, lambda: self._wgrad(ga2, a1, rts[1], wb), ga2, a1)
            ga1 = self._dgrad(ga2, rts[1], hw_mid, bits=b1)
            if blk.down:
                _off_path(side, lambda: self._wgrad(g, x, rts[3], wb), g, x)
            if bwd_pair and not first:
                deferred = (ga1, x, rts)
            else:
                _off_path(side, lambda: self._wgrad(ga1, x, rts[0], wb), ga1, x)
            if wb is not None:
                _off_path(side, wb.flush)
            pair = self._chain_pair(bi - 1) if chain and not first and not blk.down else None
            if first and not need_dx:
                g = None
            elif pair is not None:
                # conv1's data gradient + skip + the ReLU mask of x, then conv3's data gradient of the block BELOW + its ReLU mask: one launch
                g, ga2_ahead = K.conv_chain(ga1, rts[0].wpt, g, pair[0].wpt, bits1=xb, bits2=saved[bi - 1][5])
            else:
                skip = self._dgrad(g, rts[3], hw_in) if blk.down else g
                g = self._dgrad(ga1, rts[0], hw_in, res=skip, bits=None if first else xb)
            saved[bi] = None                              # release activations as we go
            if deferred is None:
                hooks(rts)
        if side is not None:
            side.join()                                   # the optimizer / caller sees complete gradients on its own stream
        return g



    # -- trainable BatchNorm2d in train() (MODEL.FREEZE_BN False): the same schedule with the statistics between conv and normalise pass
    @staticmethod
    def _bn_unit(x, rt, relu, res=None):
        """conv (plain store) -> batch statistics -> normalise (+ residual) (+ ReLU, + sign bits).  Returns (out, y, fin, count, bits)."""
        c = rt.spec
        synced = _bn_synced(rt.bn)
        y, sums, fin = K.conv_gemm_stats(x, rt.wp, arch.out_hw(x.shape[1], x.shape[2], c), c.k, c.stride, c.pad, c.dil, bn_pilot(rt.bn, x.device),
                                         bn=None if synced else rt.bn)
        if synced:          # the raw sums are exchanged first, then finalized over the global pixel count
            fin, count = bn_batch_statistics(y, rt.bn, (sums[0], sums[1]))
        else:
            count = y.numel() // y.shape[-1]
        if relu:
            out, bits = K.bn_apply(y, fin[0], fin[2], rt.bn.bias.detach(), res=res, relu=True, want_mask=True)
        else:
            out, bits = K.bn_apply(y, fin[0], fin[2], rt.bn.bias.detach(), res=res, relu=False), None
        return out, y, fin, count, bits

    def forward_batchnorm(self, x):
        """resnet.py:93-113 on batch statistics.  saved[i] = (x, bits(x) or None, [(a, y, fin, count, bits) of conv1, conv2, conv3, downsample])."""
        saved, xbits = [], None
        for blk, rts in self.blocks:
            u1 = self._bn_unit(x, rts[0], True)
            u2 = self._bn_unit(u1[0], rts[1], True)
            ud = self._bn_unit(x, rts[3], False) if blk.down else None
            u3 = self._bn_unit(u2[0], rts[2], True, res=ud[0] if blk.down else x)
            saved.append((x, xbits, [u1, u2, u3, ud]))
            x, xbits = u3[0], u3[4]
        return x, saved, xbits

    def backward_batchnorm(self, saved, fbits, dfeat, need_dx):
        """Hand-written backward of forward_batchnorm: per block, g (already masked by the block's output ReLU) -> bn3 -> conv3 -> bn2 (+ ReLU
        by its sign bits) -> conv2 -> bn1 -> conv1, whose data-gradient epilogue adds the skip gradient and applies the previous block's output
        mask (no separate add or mask pass); weight gradients on the side stream."""
        g = K.relu_mask(dfeat, fbits)
        store = getattr(self.convs[0].weight, "_mi_store", None)
        side = _SideStream.get(dfeat.device)

        batching = os.environ.get("MI_WGRAD_REDUCE_BATCH", "1") != "0" and K.PROFILE is None
        wb = [None]                                   # the block's slab reducers as one launch (K.WgradBatch), as in backward()

        def wgrad(dy, xin, rt):
            c = rt.spec

            def go():
                dw, acc = grad_slot(rt.weight)
                K.conv_wgrad(dy, xin, dw, c.k, c.stride, c.pad, c.dil, accumulate=acc, batch=wb[0])
            _off_path(side, go, dy, xin)

        for bi in range(len(self.blocks) - 1, -1, -1):
            blk, rts = self.blocks[bi]
            x, xb, (u1, u2, u3, ud) = saved[bi]
            first = bi == 0
            hw_in = (x.shape[1], x.shape[2])
            wb[0] = K.WgradBatch() if batching else None
            dy3 = bn_backward(g, u3[1], u3[2], u3[3], rts[2].bn)
            wgrad(dy3, u2[0], rts[2])
            ga2 = self._dgrad(dy3, rts[2], (u2[0].shape[1], u2[0].shape[2]))
            dy2 = bn_backward(ga2, u2[1], u2[2], u2[3], rts[1].bn, bits=u2[4])
            wgrad(dy2, u1[0], rts[1])
            ga1 = self._dgrad(dy2, rts[1], (u1[0].shape[1], u1[0].shape[2]))
            dy1 = bn_backward(ga1, u1[1], u1[2], u1[3], rts[0].bn, bits=u1[4])
            wgrad(dy1, x, rts[0])
            dyd = None
            if blk.down:
                dyd = bn_backward(g, ud[1], ud[2], ud[3], rts[3].bn)
                wgrad(dyd, x, rts[3])
            if first and not need_dx:
                g = None
            else:
                skip = self._dgrad(dyd, rts[3], hw_in) if blk.down else g
                g = self._dgrad(dy1, rts[0], hw_in, res=skip, bits=None if first else xb)
            saved[bi] = None
            if wb[0] is not None:
                _off_path(side, wb[0].flush)
            if store is not None and store.grad_hooks:
                lo, hi = store.span([p for rt in rts for p in (rt.weight, rt.bn.weight, rt.bn.bias)])
                with torch.cuda.stream(side.stream) if side is not None else contextlib.nullcontext():
                    for hook in store.grad_hooks:
                        hook(store, lo, hi)
        if side is not None:
            side.join()
        return g


# ------------------------------------------------------------------------------------------------ trainable BatchNorm2d (MODEL.FREEZE_BN=False)
def _bn_allreduce(bn, *tensors):
    """SyncBatchNorm semantics (train_distill.py:53): the raw per-channel sums of every rank are added; returns the number of ranks
    that contributed (1 when the layer is not synchronised)."""
    import torch.distributed as dist
    if not getattr(bn, "_mi_sync", False) or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return 1
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat)
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
    return dist.get_world_size()


def _bn_synced(bn):
    import torch.distributed as dist
    return getattr(bn, "_mi_sync", False) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def bn_batch_statistics(y, bn, sums=None):
    """BatchNorm2d in train() on the raw conv output y (NHWC bf16): ([mean | invstd | gamma * invstd | beta - mean * gamma * invstd], pixel count over
    the ranks that share the statistics).  The sums of y - pilot and (y - pilot)^2 (pilot = the running mean: identical on every rank, and close
    enough to the batch mean that var = E[d^2] - E[d]^2 loses 2-3 of fp32's 24 bits) come out of the conv's epilogue (`sums`) or from one pass over
    y; then the all-reduce of the raw sums when the layer is synchronised, and one finalize launch that also updates the running statistics as torch
    does - no host arithmetic in between."""
    C = y.shape[-1]
    pilot = bn_pilot(bn, y.device)
    s1, s2 = sums if sums is not None else K.bn_colsum2(y, pilot)
    count = (y.numel() // C) * _bn_allreduce(bn, s1, s2)
    return K.bn_finalize(s1, s2, pilot, count, bn), count


def bn_pilot(bn, device):
    return bn.running_mean if bn.running_mean is not None else torch.zeros(bn.num_features, dtype=torch.float32, device=device)


def bn_backward(g, y, fin, count, bn, bits=None):
    """dy of BatchNorm2d on batch statistics (+ the ReLU in front of g when `bits` carries its sign bits); the affine gradients - this rank's raw
    sums, what SyncBatchNorm hands DDP as well - go into the parameters' gradient slots."""
    (sb, ab), (sg, ag) = grad_slot(bn.bias), grad_slot(bn.weight)
    if ab or ag:
        dbeta, dgamma = K.bn_bwd_colsums(g, y, fin[0], fin[1], bits)
        sb.add_(dbeta)
        sg.add_(dgamma)
    else:
        dbeta, dgamma = K.bn_bwd_colsums(g, y, fin[0], fin[1], bits, out=(sb, sg))
    if _bn_synced(bn):
        dbeta, dgamma = dbeta.clone(), dgamma.clone()
        _bn_allreduce(bn, dbeta, dgamma)
    return K.bn_bwd_apply(g, y, fin[0], fin[1], bn.weight.detach(), dbeta, dgamma, count, bits)


class BnStemFn(torch.autograd.Function):
    """Stem with a trainable BatchNorm2d in train(): 7x7/2 conv (patch matrix + GEMM) -> batch statistics -> normalise + ReLU + 3x3/2 max-pool in ONE
    pass (the fused FrozenBN stem kernel with the batch affine).  Backward: pooled gradient routed by the stored argmax (already ReLU-masked), the
    BatchNorm backward, the stem weight gradient.  resnet.py:137-140, 177-180 with feature_extractor.py:37."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bn):
        y, saved = stem_conv_forward(x, weight)
        fin, count = bn_batch_statistics(y, bn)
        pool, idx = K.stem_pool_fwd(y, fin[2], fin[3])
        ctx.bn, ctx.count, ctx.nsaved = bn, count, len(saved)
        ctx.save_for_backward(y, fin, idx, *saved)
        return pool

    @staticmethod
    def backward(ctx, dpool):
        y, fin, idx = ctx.saved_tensors[:3]
        ones = torch.ones(y.shape[-1], dtype=torch.float32, device=y.device)
        g = K.stem_pool_bwd(dpool.contiguous(), idx, ones, (y.shape[1], y.shape[2]))          # d loss / d relu(bn(y))
        dy = bn_backward(g, y, fin, ctx.count, ctx.bn)
        return None, stem_conv_wgrad(dy, ctx.saved_tensors[3:]), None, None, None


class BnStagesFn(torch.autograd.Function):
    """layer1..layer4 with trainable BatchNorm2d on batch statistics as ONE autograd node: StageEngine.forward_batchnorm / backward_batchnorm."""

    @staticmethod
    def forward(ctx, x, eng, *params):
        eng.prepare(True)
        feat, saved, fbits = eng.forward_batchnorm(x)
        ctx.eng, ctx.saved, ctx.fbits = eng, saved, fbits
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        dx = ctx.eng.backward_batchnorm(ctx.saved, ctx.fbits, dfeat.contiguous(), ctx.needs_input_grad[0])
        ctx.saved = ctx.fbits = None
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)

# ------------------------------------------------------------------------------------------------ exact-fp32 evaluation
class Fp32Backbone:
    """Forward-only fp32 schedule of stem + layer1..layer4 for evaluation (test.py / ASPPTester): fp32 NHWC activations,
    fp32 weights, f32 MFMA (csrc/igemm_f32.hip).  Same graph and the same order of rounded operations as the reference's
    eager fp32 forward: conv -> x*scale -> +shift -> (+identity) -> relu (resnet.py:93-113, layers.py:18-23)."""

    def __init__(self, owner, plan):
        self.owner, self.plan = owner, plan
        self._sig = None
        self._packs = {}

    def _signature(self):
        bb = self.owner.backbone
        ps = list(self.owner.parameters()) + list(self.owner.buffers())
        store = getattr(bb.conv1.weight, "_mi_store", None)
        return (store.generation if store is not None else -1, sum(t._version for t in ps), bb.conv1.weight.data_ptr())

    def prepare(self):
        sig = self._signature()
        if sig == self._sig:
            return
        bb = self.owner.backbone
        self._packs = {}

        fold = bn_fold

        self._stem = (bb.conv1.weight.detach().contiguous(),) + fold(bb.bn1)
        for blk in self.plan:
            for c in arch.block_convs(blk):
                w = arch.node_at(bb, c.key).weight.detach().contiguous()
                self._packs[c.key] = (K.pack_weight_f32(w),) + fold(arch.node_at(bb, c.bn))
        self._sig = sig

    def _conv(self, x, c, relu, res=None):
        wp, sc, sh = self._packs[c.key]
        return K.conv_f32(x, wp, arch.out_hw(x.shape[1], x.shape[2], c), c.k, c.stride, c.pad, c.dil, scale=sc, bias=sh, res=res, relu=relu)

    def forward(self, x):
        """x [B,3,H,W] fp32 NCHW -> layer4 map [B,h,w,2048] fp32 NHWC."""
        self.prepare()
        w, sc, sh = self._stem
        y = K.maxpool_f32(K.stem_f32(x.contiguous(), w, sc, sh))
        for blk in self.plan:
            c1, c2, c3 = arch.block_convs(blk)[:3]
            a1 = self._conv(y, c1, True)
            a2 = self._conv(a1, c2, True)
            idn = self._conv(y, arch.block_convs(blk)[3], False) if blk.down else y
            y = self._conv(a2, c3, True, res=idn)
        return y


class Fp32Aspp:
    """classifier.py:26-29 in fp32: out = conv_0(x); out += conv_i(x), each conv with its bias, as four launches of the f32
    implicit GEMM chained through the residual input (same association as the reference's left-to-right sum)."""

    def __init__(self, owner, rates):
        self.owner, self.rates = owner, rates
        self._sig = None

    def prepare(self):
        convs = [getattr(self.owner.conv2d_list, str(i)) for i in range(4)]
        store = getattr(convs[0].weight, "_mi_store", None)
        sig = (store.generation if store is not None else -1, sum(c.weight._version + c.bias._version for c in convs), convs[0].weight.data_ptr())
        if sig != self._sig:
            self._wp = [K.pack_weight_f32(c.weight.detach().contiguous()) for c in convs]
            self._b = [c.bias.detach().contiguous() for c in convs]
            self._sig = sig

    def forward(self, x):
        """x [B,h,w,C] fp32 NHWC -> low [B,h,w,K] fp32 NHWC."""
        self.prepare()
        h, w = x.shape[1], x.shape[2]
        out = None
        for wp, b, d in zip(self._wp, self._b, self.rates):
            out = K.conv_f32(x, wp, (h, w), 3, 1, d, d, bias=b, res=out)
        return out


def stem_conv_forward(x, weight):
    """x [B,3,H,W] bf16 channels_last, weight fp32 [64,3,7,7] -> (y [B,Hc,Wc,64] bf16 NHWC, what the weight gradient needs): patch matrix + plain
    GEMM on the implicit-GEMM kernel - no library convolution anywhere in the step (rounds 1-4 kept an MIOpen branch behind MI_STEM_CONV; removed)."""
    y, col = K.stem_conv_fwd(x, weight)
    return y, (col,)


def stem_conv_wgrad(dy_nhwc, saved):
    """d loss / d conv1.weight: the 1x1 weight-gradient kernel on the forward's patch matrix (bit-reproducible)."""
    return K.stem_wgrad(dy_nhwc, col=saved[0])


class StemConvFn(torch.autograd.Function):
    """The 7x7/2 stem conv alone (the trainable-BatchNorm stem); the input image needs no gradient.
    x [B,3,H,W] bf16 channels_last, weight fp32 -> [B,Hc,Wc,64] bf16 NHWC."""

    @staticmethod
    def forward(ctx, x, weight):
        y, saved = stem_conv_forward(x, weight)
        ctx.save_for_backward(*saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        return None, stem_conv_wgrad(dy.contiguous(), ctx.saved_tensors)


class StemFn(torch.autograd.Function):
    """Stem: 7x7/2 conv (patch matrix + GEMM) + fused FrozenBN/ReLU/max-pool HIP kernel.
    x [B,3,H,W] bf16 channels_last, weight fp32 [64,3,7,7] -> pooled [B,Hp,Wp,64] bf16 NHWC."""

    @staticmethod
    def forward(ctx, x, weight, scale, shift):
        y, saved = stem_conv_forward(x, weight)
        pool, idx = K.stem_pool_fwd(y, scale, shift)
        ctx.save_for_backward(idx, scale, *saved)
        ctx.conv_hw = (y.shape[1], y.shape[2])
        return pool

    @staticmethod
    def backward(ctx, dpool):
        idx, scale = ctx.saved_tensors[:2]
        dy = K.stem_pool_bwd(dpool.contiguous(), idx, scale, ctx.conv_hw)                      # [B,Hc,Wc,64] bf16 NHWC
        return None, stem_conv_wgrad(dy, ctx.saved_tensors[2:]), None, None


class StagesFn(torch.autograd.Function):
    """pooled stem output [B,H,W,64] bf16 NHWC -> layer4 feature [B,h,w,2048] bf16 NHWC."""

    @staticmethod
    def forward(ctx, x, eng, *weights):
        train = any(ctx.needs_input_grad)            # (grad mode is off inside Function.forward)
        eng.prepare(train)
        feat, saved, fbits = eng.forward(x, save=train)
        ctx.eng, ctx.saved, ctx.fbits = eng, saved, fbits
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        dfeat = dfeat.contiguous()
        dx = ctx.eng.backward(ctx.saved, ctx.fbits, dfeat, ctx.needs_input_grad[0])
        ctx.saved = ctx.fbits = None
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


# ------------------------------------------------------------------------------------------------ ASPP head
class AsppEngine:
    """Four dilated 3x3 convs (2048 -> K) summed, as ONE plain GEMM over all rates x taps x classes plus a
    col2im of the tiny K-channel planes; backward as im2col + two plain GEMMs."""

    def __init__(self, owner, rates, num_classes, in_channels):
        self.owner, self.rates, self.K, self.C = owner, tuple(int(r) for r in rates), num_classes, in_channels
        self.wall = self.wallT = None
        self._sig = None
        self._have_dgrad = False

    def _w4(self):
        ws = [getattr(self.owner.conv2d_list, str(i)).weight for i in range(4)]
        n = ws[0].numel()
        base = ws[0].data_ptr()
        if all(w.data_ptr() == base + 4 * n * i for i, w in enumerate(ws)):
            return torch.as_strided(ws[0].detach(), (4,) + tuple(ws[0].shape), (n,) + tuple(ws[0].stride())), ws
        return torch.stack([w.detach() for w in ws]).contiguous(), ws

    def _b4(self):
        bs = [getattr(self.owner.conv2d_list, str(i)).bias for i in range(4)]
        base = bs[0].data_ptr()
        if all(b.data_ptr() == base + 4 * self.K * i for i, b in enumerate(bs)):
            return torch.as_strided(bs[0].detach(), (4, self.K), (self.K, 1)), bs
        return torch.stack([b.detach() for b in bs]).contiguous(), bs

    def prepare(self, train):
        w4, ws = self._w4()
        store = getattr(ws[0], "_mi_store", None)
        sig = (store.generation if store is not None else -1, sum(w._version for w in ws), ws[0].data_ptr())
        if sig != self._sig or (train and not self._have_dgrad):
            self.wall = K.aspp_pack_fwd(w4, out=self.wall)
            if train:
                self.wallT = K.aspp_pack_dgrad(w4, out=self.wallT)
            self._sig = sig
            self._have_dgrad = train

    def forward(self, x):
        """x [B,h,w,C] bf16 -> low [B,h,w,K] fp32 (1/8-resolution logits, classifier.py:27-29)."""
        B, h, w, _ = x.shape
        z = K.conv_gemm(x, self.wall, (h, w), zsplit=K.ASPP_ZGW, flop_cols=self.K)
        b4, _ = self._b4()
        return K.aspp_col2im(z, b4, B, h, w, self.K, self.rates)

    def backward(self, x, dlow, need_dx, msk=None):
        """dlow [B,h,w,K] fp32 = d loss / d low.  Writes weight/bias grads, returns d loss / d x (bf16)."""
        h, w = x.shape[1], x.shape[2]
        g = K.aspp_im2col(dlow, self.rates)
        w4, ws = self._w4()
        b4, bs = self._b4()
        side = _SideStream.get(x.device)

        def weight_and_bias_grads():
            slots = [grad_slot(p) for p in ws]
            n = ws[0].numel()
            base = slots[0][0].data_ptr()
            stacked = all(s[0].data_ptr() == base + 4 * n * i for i, s in enumerate(slots)) and len({s[1] for s in slots}) == 1
            if stacked:
                dw4 = torch.as_strided(slots[0][0], (4,) + tuple(ws[0].shape), (n,) + tuple(ws[0].stride()))
                K.conv_wgrad(g, x, dw4, out_map=1, ncls=self.K, accumulate=slots[0][1])
            else:
                dw4 = torch.empty((4,) + tuple(ws[0].shape), dtype=torch.float32, device=x.device)
                K.conv_wgrad(g, x, dw4, out_map=1, ncls=self.K)
                for i, (slot, acc) in enumerate(slots):
                    slot.add_(dw4[i]) if acc else slot.copy_(dw4[i])
            bslots = [grad_slot(p) for p in bs]
            bbase = bslots[0][0].data_ptr()
            if all(s[0].data_ptr() == bbase + 4 * self.K * i for i, s in enumerate(bslots)) and len({s[1] for s in bslots}) == 1:
                K.aspp_bias_grad(dlow, torch.as_strided(bslots[0][0], (4, self.K), (self.K, 1)), accumulate=bslots[0][1])
            else:
                db4 = torch.empty((4, self.K), dtype=torch.float32, device=x.device)
                K.aspp_bias_grad(dlow, db4)
                for i, (slot, acc) in enumerate(bslots):
                    slot.add_(db4[i]) if acc else slot.copy_(db4[i])

        # off the critical path: runs beside the data-gradient GEMM below; joined right after it is enqueued
        _off_path(side, weight_and_bias_grads, g, x, dlow)
        dx = K.conv_gemm(g, self.wallT, (h, w), msk=msk, flop_cols=36 * self.K) if need_dx else None
        if side is not None:
            side.join()
        store = getattr(ws[0], "_mi_store", None)
        if store is not None and store.grad_hooks:
            for hook in store.grad_hooks:
                hook(store, 0, store.total)
        return dx


class AsppFn(torch.autograd.Function):
    """feature [B,h,w,C] bf16 NHWC -> low-resolution logits [B,h,w,K] fp32 NHWC."""

    @staticmethod
    def forward(ctx, x, eng, *params):
        train = any(ctx.needs_input_grad)
        eng.prepare(train)
        ctx.eng, ctx.x = eng, (x if train else None)
        return eng.forward(x)

    @staticmethod
    def backward(ctx, dlow):
        dx = ctx.eng.backward(ctx.x, dlow.contiguous().float(), ctx.needs_input_grad[0])
        ctx.x = None
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class AsppLossFn(torch.autograd.Function):
    """Training fast path: ASPP head + bilinear upsample to the label size + CrossEntropyLoss(ignore_index),
    i.e. reference aspp_trainer.py:89-91 `criterion(classifier(feat, size), label)`, without materialising the
    [B,K,H,W] logits.  The loss gradient w.r.t. the low-resolution logits is produced in the same pass."""

    @staticmethod
    def forward(ctx, x, labels, eng, ignore_index, temperature, *params):
        train = any(ctx.needs_input_grad)
        eng.prepare(train)
        low = eng.forward(x)
        eng.last_low = low                   # FADA derives its soft labels from these (aspp_fada.py:85-87)
        if temperature != 1.0:               # criterion(pred.div(T), label): bilinear upsampling commutes with the scaling
            loss_out, dlow = K.upsample_ce(low * (1.0 / temperature), labels, want_grad=train, grad_scale=1.0 / temperature,
                                           ignore_index=ignore_index)
        else:
            loss_out, dlow = K.upsample_ce(low, labels, want_grad=train, ignore_index=ignore_index)
        ctx.eng, ctx.x, ctx.dlow = eng, (x if train else None), dlow
        ctx.loss_out = eng.last_loss_out = loss_out          # [loss, n_valid, out-of-range labels, -]: see K.check_labels
        # every step's count goes into a persistent device counter (the kernel zeroes its own per call): the trainer reads it once per logging
        # window, so a bad label in ANY step of the window raises, as torch's device assert would - also under HIP-graph replay (the add is captured)
        bad = getattr(eng, "bad_labels", None)
        if bad is None or bad.device != loss_out.device:
            bad = eng.bad_labels = torch.zeros(1, dtype=torch.float32, device=loss_out.device)
        bad.add_(loss_out[2:3])
        return loss_out[0].clone()

    @staticmethod
    def backward(ctx, gout):
        dlow = ctx.dlow * gout
        dx = ctx.eng.backward(ctx.x, dlow, ctx.needs_input_grad[0])
        ctx.x = ctx.dlow = None
        return (dx, None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 5)


class UpsampleFn(torch.autograd.Function):
    """low [B,h,w,K] fp32 NHWC -> [B,K,H,W] fp32 NCHW, bilinear align_corners=True (classifier.py:31)."""

    @staticmethod
    def forward(ctx, low, size):
        ctx.hw = (low.shape[1], low.shape[2])
        return K.upsample_ac_fwd(low, tuple(int(s) for s in size))

    @staticmethod
    def backward(ctx, dup):
        return K.upsample_ac_bwd(dup.contiguous(), ctx.hw), None


class SoftmaxCEFn(torch.autograd.Function):
    """CrossEntropyLoss(ignore_index) on materialised [B,K,H,W] fp32 logits (aspp_trainer.py:61,91)."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        logits = logits.contiguous()
        out = K.softmax_ce_fwd(logits, labels, ignore_index)
        K.check_labels(out, logits.shape[1], "CrossEntropyLoss target")      # torch raises for these; this (unfused, API-parity) path syncs
        ctx.save_for_backward(logits, labels, out)
        ctx.ignore_index = ignore_index
        return out[0].clone()

    @staticmethod
    def backward(ctx, gout):
        logits, labels, out = ctx.saved_tensors
        d = K.softmax_ce_bwd(logits, labels, out, 1.0, ctx.ignore_index)
        return d * gout, None, None
