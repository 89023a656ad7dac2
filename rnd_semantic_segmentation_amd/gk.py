"""Tensor-level wrappers over the view-based C-ABI entry points of the PraNet path (include/mi355seg.h: mi_gconv ... mi_gra_bwd).

A tensor handed to these functions is an NHWC tensor [B,H,W,C] or a CHANNEL SLICE of one (`t[..., a:b]`): last-dim stride 1, pixel
stride `ld` (elements per pixel row of the parent), no other gaps.  One-channel fp32 maps are [B,H,W,1] with ld 1.
"""
import ctypes

import torch

from . import _lib, kernels as _K
from ._lib import check
from .kernels import _p, _stream, _workspace, GATHER_FWD, GATHER_DGRAD  # noqa: F401


class _TimedLib:
    """kernels.PROFILE (bench.py's instrumented steps): every entry point called through this proxy is bracketed by HIP events on the launch stream
    and recorded under its own name, with the algorithmic FLOPs the wrapper announced in `_work` (0 for the elementwise / reduction kernels)."""

    def __getattr__(self, name):
        fn = getattr(_lib.lib(), name)
        if name.endswith(("_workspace", "_elems", "_bytes")):
            return fn

        def call(*args):
            work, tag = _work[0], _work[1]
            _work[0], _work[1] = 0.0, None
            return _K._timed(name[3:], work, lambda: fn(*args), tag)
        return call


_work = [0.0, None]
_timed_lib = _TimedLib()


def _L():
    return _timed_lib if _K.PROFILE is not None else _lib.lib()

OP_ADD, OP_MUL, OP_COPY, OP_RELU_MASK, OP_MULRELU = 0, 1, 2, 3, 4


def view(t, dtype=None):
    """(pointer, ld) of an NHWC tensor or channel slice; raises for anything the kernels cannot address."""
    if not t.is_cuda:
        raise _lib.MiError("tensor must live on the GPU (no CPU path exists)")
    if dtype is not None and t.dtype != dtype:
        raise _lib.MiError("expected %s, got %s" % (dtype, t.dtype))
    if t.dim() != 4:
        raise _lib.MiError("expected [B,H,W,C], got %s" % (tuple(t.shape),))
    B, H, W, C = t.shape
    s = t.stride()
    ld = s[2] if W > 1 else (s[1] // max(W, 1) if H > 1 else (s[0] // max(H * W, 1) if B > 1 else C))
    if C > 1 and s[3] != 1:
        raise _lib.MiError("channels must be contiguous, strides %s" % (s,))
    if ld < C or (W > 1 and s[2] != ld) or (H > 1 and s[1] != W * ld) or (B > 1 and s[0] != H * W * ld):
        raise _lib.MiError("not a channel-slice view of an NHWC tensor: shape %s strides %s" % (tuple(t.shape), s))
    return ctypes.c_void_p(t.data_ptr()), int(ld)


def new(B, H, W, C, device, dtype=torch.bfloat16):
    return torch.empty((B, H, W, C), dtype=dtype, device=device)


def conv_out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw):
    return (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1, (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1


def gconv(a, wp, N, geom, out=None, mode=GATHER_FWD, out_hw=None, bias=None, stats=False, out_f32=False):
    """geom = (kh, kw, sh, sw, ph, pw, dh, dw).  FWD: a is the input, returns (out [B,Ho,Wo,N], stats or None).
    DGRAD: a is d loss / d conv output, out_hw the forward input's (H, W), wp the transposed pack, N the forward input channels."""
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    B, Ha, Wa, Ca = a.shape
    if mode == GATHER_FWD:
        Ho, Wo = conv_out_hw(Ha, Wa, *geom)
    else:
        Ho, Wo = out_hw
    if out is None:
        out = new(B, Ho, Wo, N, a.device, torch.float32 if out_f32 else torch.bfloat16)
    pa, lda = view(a, torch.bfloat16)
    po, ldo = view(out, torch.float32 if out_f32 else torch.bfloat16)
    if tuple(out.shape) != (B, Ho, Wo, N):
        raise _lib.MiError("gconv: out is %s, the conv writes %s" % (tuple(out.shape), (B, Ho, Wo, N)))
    L = _L()
    st = None
    if stats:
        st = torch.empty(int(L.mi_gconv_stats_elems(B, Ho, Wo, N)), dtype=torch.float32, device=a.device)
    if _K.PROFILE is not None:
        _work[0], _work[1] = 2.0 * B * Ho * Wo * N * Ca * kh * kw, ("gconv", kh, kw, Ca, N, B * Ho * Wo, mode)          # 2 * pixels * C_out * C_in * taps
    check(L.mi_gconv(pa, lda, _p(wp), po, ldo, B, Ha, Wa, Ca, Ho, Wo, N, kh, kw, sh, sw, ph, pw, dh, dw, mode, _p(bias), _p(st), int(out_f32), _stream()),
          "mi_gconv")
    return out, st


_tickets = {}
# second-level reductions inside the first launch where they are small: OFF by default since round 5 (measured neutral under graph replay and in GALD's
# eager step; they pay only in eager PraNet, which is not the trainer's default): MI_INLAUNCH=1 turns them on (same bits either way)
INLAUNCH = __import__("os").environ.get("MI_INLAUNCH", "0") == "1"


def tickets(device, n=256):
    """Zeroed 32-bit words for the in-launch reductions (mi_common.h: mi_last_arriver), one buffer per (device, stream): launches on one stream
    are ordered and every launch leaves its words zero; launches on different streams must not share them."""
    key = (device.index, _stream().value)
    t = _tickets.get(key)
    if t is None or t.numel() < n:
        t = _tickets[key] = torch.zeros(max(n, 256 + 4096), dtype=torch.int32, device=device)
    return t


def _check_tickets(rc, what, device):
    """check() for the launches that use the ticket words: a launch that failed may have left words non-zero (some workgroups drew tickets, the last
    arriver never reset them); the next launch on the stream would then find no 'last arriver' and leave its second-level results unwritten, silently.
    Drop the stream's buffer: the next call allocates a zeroed one."""
    if rc != 0:
        _tickets.pop((device.index, _stream().value), None)
    check(rc, what)


_inlaunch_px = [None]


def gconv_bn_fits(B, Ho, Wo):
    if _inlaunch_px[0] is None:
        _inlaunch_px[0] = int(_lib.lib().mi_gconv_bn_inlaunch_max_pixels())
    return B * Ho * Wo <= _inlaunch_px[0]


def gconv_bn(a, wp, N, geom, bn_weight, bn_bias, running_mean, running_var, momentum, eps, bias=None, out=None):
    """Forward conv + BatchNorm2d statistics AND finalize in one launch (small maps: gconv_bn_fits) -> (y [B,Ho,Wo,N] bf16, fin [4,N]: mean, invstd,
    scale, shift); the running statistics are updated as torch does."""
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    B, Ha, Wa, Ca = a.shape
    Ho, Wo = conv_out_hw(Ha, Wa, *geom)
    if out is None:
        out = new(B, Ho, Wo, N, a.device)
    pa, lda = view(a, torch.bfloat16)
    po, ldo = view(out, torch.bfloat16)
    L = _L()
    st = torch.empty(int(L.mi_gconv_stats_elems(B, Ho, Wo, N)), dtype=torch.float32, device=a.device)
    fin = torch.empty((4, N), dtype=torch.float32, device=a.device)
    tk = tickets(a.device, (N + 31) // 32)
    if _K.PROFILE is not None:
        _work[0], _work[1] = 2.0 * B * Ho * Wo * N * Ca * kh * kw, ("gconv", kh, kw, Ca, N, B * Ho * Wo, GATHER_FWD)
    _check_tickets(L.mi_gconv_bn(pa, lda, _p(wp), po, ldo, B, Ha, Wa, Ca, Ho, Wo, N, kh, kw, sh, sw, ph, pw, dh, dw, _p(bias), _p(st), _p(tk), _p(bn_weight), _p(bn_bias),
                                 _p(running_mean), _p(running_var), float(momentum), float(eps), _p(fin), _stream()), "mi_gconv_bn", a.device)
    return out, fin


def gconv_wgrad(dy, x, dw, geom, accumulate=False):
    """dw [O,I,kh,kw] fp32 (+)= conv weight gradient; dy [B,Ho,Wo,O], x [B,Ha,Wa,I] bf16 views."""
    kh, kw, sh, sw, ph, pw, dh, dw_ = geom
    B, Ho, Wo, O = dy.shape
    _, Ha, Wa, I = x.shape
    if not (dw.is_contiguous() and dw.dtype == torch.float32 and dw.numel() == O * I * kh * kw):
        raise _lib.MiError("gconv_wgrad: dw must be contiguous fp32 [O,I,kh,kw]")
    py, ldy = view(dy, torch.bfloat16)
    px, ldx = view(x, torch.bfloat16)
    L = _L()
    ws = _workspace(L.mi_gconv_wgrad_workspace(B, Ho, Wo, O, I, kh, kw), dy.device, "gwgrad")
    if _K.PROFILE is not None:
        _work[0], _work[1] = 2.0 * B * Ho * Wo * O * I * kh * kw, ("gwgrad", kh, kw, I, O, B * Ho * Wo, 0)
    tk = tickets(dy.device, 256 + 4096)[256:] if INLAUNCH else None          # (words 0 .. 255: the conv / column-sum reductions)
    _check_tickets(L.mi_gconv_wgrad(py, ldy, px, ldx, _p(dw), B, Ha, Wa, I, Ho, Wo, O, kh, kw, sh, sw, ph, pw, dh, dw_, int(accumulate), _p(ws), ws.numel(), _p(tk),
                                    tk.numel() if tk is not None else 0, _stream()), "mi_gconv_wgrad", dy.device)
    return dw


class _WgradJob(ctypes.Structure):
    """include/mi355seg.h: MiWgradJob"""
    _fields_ = ([("dy", ctypes.c_void_p), ("ldy", ctypes.c_long), ("x", ctypes.c_void_p), ("ldx", ctypes.c_long), ("dw", ctypes.c_void_p)] +
                [(_k, ctypes.c_int) for _k in ("B", "Ha", "Wa", "I", "Ho", "Wo", "O", "kh", "kw", "sh", "sw", "ph", "pw", "dh", "dw_", "accumulate")])


def gconv_wgrad_multi(jobs):
    """The weight gradients of many convs in one go: jobs = [(dy, x, dw, geom, accumulate), ...] with gconv_wgrad's operands; no two jobs may share a dw."""
    if not jobs:
        return
    n = len(jobs)
    arr = (_WgradJob * n)()
    flops = 0.0
    for j, (dy, x, dw, geom, accumulate) in zip(arr, jobs):
        kh, kw, sh, sw, ph, pw, dh, dw_ = geom
        B, Ho, Wo, O = dy.shape
        _, Ha, Wa, I = x.shape
        if not (dw.is_contiguous() and dw.dtype == torch.float32 and dw.numel() == O * I * kh * kw):
            raise _lib.MiError("gconv_wgrad_multi: dw must be contiguous fp32 [O,I,kh,kw]")
        py, ldy = view(dy, torch.bfloat16)
        px, ldx = view(x, torch.bfloat16)
        j.dy, j.ldy, j.x, j.ldx, j.dw = py.value if hasattr(py, "value") else py, ldy, px.value if hasattr(px, "value") else px, ldx, dw.data_ptr()
        j.B, j.Ha, j.Wa, j.I, j.Ho, j.Wo, j.O = B, Ha, Wa, I, Ho, Wo, O
        j.kh, j.kw, j.sh, j.sw, j.ph, j.pw, j.dh, j.dw_, j.accumulate = kh, kw, sh, sw, ph, pw, dh, dw_, int(accumulate)
        flops += 2.0 * B * Ho * Wo * O * I * kh * kw
    dev = jobs[0][0].device
    L = _L()
    pj = ctypes.cast(arr, ctypes.c_void_p)
    need = L.mi_gconv_wgrad_multi_workspace(pj, n)
    if not need:
        raise _lib.MiError("gconv_wgrad_multi: a job has an empty or oversized shape")
    ws = _workspace(need, dev, "gwgrad_multi")
    table = _workspace(L.mi_gconv_wgrad_multi_table_bytes(n), dev, "gwgrad_table")
    if _K.PROFILE is not None:
        _work[0], _work[1] = flops, ("gwgrad_multi", n, 0, 0, 0, 0, 0)
    check(L.mi_gconv_wgrad_multi(pj, n, _p(table), table.numel(), _p(ws), ws.numel(), _stream()), "mi_gconv_wgrad_multi")


def gconv_pack_multi(wflat, wp, wpt, table_dev, n_desc, total_blocks):
    check(_L().mi_gconv_pack_multi(_p(wflat), _p(wp), _p(wpt), _p(table_dev), n_desc, total_blocks, _stream()), "mi_gconv_pack_multi")


def pack_elems(O, I, kh, kw):
    return int(_L().mi_gconv_pack_elems(O, I, kh, kw))


def gconv_pack(w):
    """One conv's fp32 OIHW weight -> (forward pack, data-gradient pack); the module-level path packs every conv in one launch."""
    O, I, kh, kw = w.shape
    n = pack_elems(O, I, kh, kw)
    wp = torch.empty(n, dtype=torch.bfloat16, device=w.device)
    wpt = torch.empty(n, dtype=torch.bfloat16, device=w.device)
    table = torch.tensor([[0, 0, 0, O, I, kh * kw, 0, 0]], dtype=torch.int64, device=w.device)
    gconv_pack_multi(w.detach().contiguous().view(-1), wp, wpt, table, 1, -(-n // 1024))
    return wp, wpt


def gbn_finalize(stats, C, count, gamma, beta, running_mean, running_var, momentum, eps):
    dev = stats.device
    out = torch.empty((4, C), dtype=torch.float32, device=dev)       # mean, invstd, scale, shift
    tiles = stats.numel() // (2 * C)
    check(_L().mi_gbn_finalize(_p(stats), tiles, C, int(count), _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(momentum), float(eps),
                                     _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), _stream()), "mi_gbn_finalize")
    return out


def gbn_fold(gamma, beta, rm, rv, eps):
    C = gamma.numel()
    out = torch.empty((2, C), dtype=torch.float32, device=gamma.device)
    check(_L().mi_gbn_fold(_p(gamma), _p(beta), _p(rm), _p(rv), float(eps), _p(out[0]), _p(out[1]), C, _stream()), "mi_gbn_fold")
    return out[0], out[1]


def gbn_apply(y, scale, shift, relu, add=None, out=None, out_f32=False):
    B, H, W, C = y.shape
    if out is None:
        out = new(B, H, W, C, y.device, torch.float32 if out_f32 else torch.bfloat16)
    py, ldy = view(y, torch.bfloat16)
    po, ldo = view(out, torch.float32 if out_f32 else torch.bfloat16)
    pa, lda = view(add, torch.bfloat16) if add is not None else (None, 0)
    check(_L().mi_gbn_apply(py, ldy, _p(scale), _p(shift), pa, lda, po, ldo, int(out_f32), B * H * W, C, int(relu), _stream()), "mi_gbn_apply")
    return out


APPLY_MAX_EXTRA = 4


def gbn_apply_multi(y, scale, shift, relu, extras, add=None, out=None):
    """gbn_apply (bf16) whose result also goes to other views: extras = [(c0, c1, dst, add2 or None), ...] (at most APPLY_MAX_EXTRA) - channels [c0, c1) of
    the stored result (+ add2) into dst, what a copy / add kernel reading `out` back would write, bit for bit.  Returns out."""
    B, H, W, C = y.shape
    if out is None:
        out = new(B, H, W, C, y.device)
    n = len(extras)
    if n > APPLY_MAX_EXTRA:
        raise _lib.MiError("gbn_apply_multi: %d extra destinations (at most %d)" % (n, APPLY_MAX_EXTRA))
    py, ldy = view(y, torch.bfloat16)
    po, ldo = view(out, torch.bfloat16)
    pa, lda = view(add, torch.bfloat16) if add is not None else (None, 0)
    c0 = (ctypes.c_int * APPLY_MAX_EXTRA)()
    c1 = (ctypes.c_int * APPLY_MAX_EXTRA)()
    dst = (ctypes.c_void_p * APPLY_MAX_EXTRA)()
    ldd = (ctypes.c_long * APPLY_MAX_EXTRA)()
    add2 = (ctypes.c_void_p * APPLY_MAX_EXTRA)()
    lda2 = (ctypes.c_long * APPLY_MAX_EXTRA)()
    for k, (a0, a1, d, a2) in enumerate(extras):
        if tuple(d.shape[:3]) != (B, H, W) or d.shape[-1] != a1 - a0 or (a2 is not None and tuple(a2.shape) != tuple(d.shape)):
            raise _lib.MiError("gbn_apply_multi: extra destination %d is %s for channels [%d, %d) of %s" % (k, tuple(d.shape), a0, a1, tuple(y.shape)))
        pd, l_d = view(d, torch.bfloat16)
        c0[k], c1[k], dst[k], ldd[k] = int(a0), int(a1), pd.value, l_d
        if a2 is not None:
            p2, l2 = view(a2, torch.bfloat16)
            add2[k], lda2[k] = p2.value, l2
    check(_L().mi_gbn_apply_multi(py, ldy, _p(scale), _p(shift), pa, lda, po, ldo, B * H * W, C, int(relu), n, c0, c1, dst, ldd, add2, lda2, _stream()),
          "mi_gbn_apply_multi")
    return out


def _gm(g, mask, relu6=False):
    pg, ldg = view(g)
    gf = g.dtype == torch.float32
    if mask is None:
        return pg, ldg, int(gf), None, 0, 0
    pm, ldm = view(mask)
    return pg, ldg, int(gf), pm, ldm, int(mask.dtype == torch.float32) | (2 if relu6 else 0)      # flag word: bit 0 fp32 mask, bit 1 ReLU6 mask


def gbn_bwd_sums(g, y, mask, mean, invstd, dbeta, dgamma, accumulate=False, relu6=False):
    """dbeta / dgamma: fp32 [C] slots written (or accumulated into); y None: dbeta only.  relu6: the mask is a ReLU6 output."""
    B, H, W, C = g.shape
    M = B * H * W
    pg, ldg, gf, pm, ldm, mf = _gm(g, mask, relu6)
    py, ldy = view(y, torch.bfloat16) if y is not None else (None, 0)
    L = _L()
    ws = _workspace(L.mi_gcolsum_workspace(M, C), g.device, "gcolsum")
    check(L.mi_gbn_bwd_sums(pg, ldg, gf, py, ldy, pm, ldm, mf, _p(mean), _p(invstd), M, C, _p(dbeta), _p(dgamma), int(accumulate), _p(ws), ws.numel(),
                            _stream()), "mi_gbn_bwd_sums")


def gbn_bwd_apply(g, y, mask, mean, invstd, gamma, dbeta, dgamma, count, out=None, relu6=False):
    B, H, W, C = g.shape
    if out is None:
        out = new(B, H, W, C, g.device)
    pg, ldg, gf, pm, ldm, mf = _gm(g, mask, relu6)
    py, ldy = view(y, torch.bfloat16)
    po, ldo = view(out, torch.bfloat16)
    check(_L().mi_gbn_bwd_apply(pg, ldg, gf, py, ldy, pm, ldm, mf, _p(mean), _p(invstd), _p(gamma), _p(dbeta), _p(dgamma), ctypes.c_float(1.0 / count),
                                      po, ldo, B * H * W, C, _stream()), "mi_gbn_bwd_apply")
    return out


def gbinary(op, a, b=None, out=None, out_dtype=None):
    B, H, W, C = a.shape
    if out_dtype is None:
        out_dtype = a.dtype if out is None else out.dtype
    if out is None:
        out = new(B, H, W, C, a.device, out_dtype)
    if a.dtype == out.dtype:
        dt = 0 if a.dtype == torch.bfloat16 else 1
    else:
        dt = 2 if a.dtype == torch.float32 else 3
    pa, lda = view(a)
    pb, ldb = view(b, a.dtype) if b is not None else (None, 0)
    po, ldo = view(out)
    check(_L().mi_gbinary(op, dt, pa, lda, pb, ldb, po, ldo, B * H * W, C, _stream()), "mi_gbinary")
    return out


def gavgpool(x, k, stride, pad, include_pad, out_hw, out=None):
    B, H, W, C = x.shape
    Ho, Wo = out_hw
    if out is None:
        out = new(B, Ho, Wo, C, x.device)
    px, ldx = view(x, torch.bfloat16)
    po, ldo = view(out, torch.bfloat16)
    check(_L().mi_gavgpool(px, ldx, po, ldo, B, H, W, C, Ho, Wo, k, stride, pad, int(include_pad), 0, _stream()), "mi_gavgpool")
    return out


def gavgpool_bwd(dout, in_hw, k, stride, pad, include_pad, dx=None):
    B, Ho, Wo, C = dout.shape
    H, W = in_hw
    if dx is None:
        dx = new(B, H, W, C, dout.device)
    px, ldx = view(dx, torch.bfloat16)
    po, ldo = view(dout, torch.bfloat16)
    check(_L().mi_gavgpool(px, ldx, po, ldo, B, H, W, C, Ho, Wo, k, stride, pad, int(include_pad), 1, _stream()), "mi_gavgpool(bwd)")
    return dx


def resize_scales(in_hw, out_hw, align_corners, scale_factor=None):
    """The per-axis source scale as ATen's area_pixel_compute_scale computes it (fp32)."""
    def one(i, o):
        if align_corners:
            return float(torch.tensor((i - 1) / (o - 1) if o > 1 else 0.0, dtype=torch.float32))
        if scale_factor is not None:
            return float(torch.tensor(1.0 / scale_factor, dtype=torch.float32))
        return float(torch.tensor(i / o, dtype=torch.float32))
    return one(in_hw[0], out_hw[0]), one(in_hw[1], out_hw[1])


def gresize(x, out_hw, align_corners, scale_factor=None, out=None):
    B, H, W, C = x.shape
    Ho, Wo = out_hw
    if out is None:
        out = new(B, Ho, Wo, C, x.device, x.dtype)
    sh, sw = resize_scales((H, W), out_hw, align_corners, scale_factor)
    px, ldx = view(x)
    po, ldo = view(out, x.dtype)
    check(_L().mi_gresize(px, ldx, po, ldo, int(x.dtype == torch.float32), B, H, W, C, Ho, Wo, int(align_corners), sh, sw, 0, _stream()), "mi_gresize")
    return out


def gresize_bwd(dout, in_hw, align_corners, scale_factor=None, dx=None):
    B, Ho, Wo, C = dout.shape
    H, W = in_hw
    if dx is None:
        dx = new(B, H, W, C, dout.device, dout.dtype)
    sh, sw = resize_scales((H, W), (Ho, Wo), align_corners, scale_factor)
    px, ldx = view(dx, dout.dtype)
    po, ldo = view(dout)
    check(_L().mi_gresize(px, ldx, po, ldo, int(dout.dtype == torch.float32), B, H, W, C, Ho, Wo, int(align_corners), sh, sw, 1, _stream()), "mi_gresize(bwd)")
    return dx


def gra_fwd(gate, feat, out=None):
    """gate fp32 [B,H,W,1], feat bf16 [B,H,W,C] -> (1 - sigmoid(gate)) * feat"""
    B, H, W, C = feat.shape
    if out is None:
        out = new(B, H, W, C, feat.device)
    pf, ldf = view(feat, torch.bfloat16)
    po, ldo = view(out, torch.bfloat16)
    if not (gate.dtype == torch.float32 and gate.is_contiguous() and gate.numel() == B * H * W):
        raise _lib.MiError("gra_fwd: gate must be contiguous fp32 with one value per pixel")
    check(_L().mi_gra_fwd(_p(gate), pf, ldf, po, ldo, B * H * W, C, _stream()), "mi_gra_fwd")
    return out


def gra_bwd(gate, feat, dy):
    B, H, W, C = feat.shape
    dfeat = new(B, H, W, C, feat.device)
    dgate = torch.empty((B, H, W, 1), dtype=torch.float32, device=feat.device)
    pf, ldf = view(feat, torch.bfloat16)
    pd, ldd = view(dy, torch.bfloat16)
    po, ldo = view(dfeat, torch.bfloat16)
    check(_L().mi_gra_bwd(_p(gate), pf, ldf, pd, ldd, po, ldo, _p(dgate), B * H * W, C, _stream()), "mi_gra_bwd")
    return dfeat, dgate


# ---------------------------------------------------------------------------------------------- GALD / GCPA path, first kernels (csrc/gald.hip)
def gdwconv(x, w, bias, stride, pad, out=None, stats=False):
    """Depthwise 3x3 conv with bias: x [B,H,W,C] bf16 view, w fp32 [C,1,3,3], bias fp32 [C] or None -> (out [B,Ho,Wo,C], tile statistics or None)."""
    B, H, W, C = x.shape
    Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    if out is None:
        out = new(B, Ho, Wo, C, x.device)
    px, ldx = view(x, torch.bfloat16)
    po, ldo = view(out, torch.bfloat16)
    L = _L()
    st = torch.empty(int(L.mi_gdwconv_stats_elems(B, Ho, Wo, C)), dtype=torch.float32, device=x.device) if stats else None
    check(L.mi_gdwconv(px, ldx, _p(w), _p(bias), po, ldo, B, H, W, C, Ho, Wo, stride, pad, _p(st), _stream()), "mi_gdwconv")
    return out, st


def gdwconv_backward(dy, x, w, dw, dbias, stride, pad, need_dx=True, accumulate=False):
    """dw [C,1,3,3] / dbias [C] fp32 slots written (or accumulated into); returns d loss / d x (bf16) or None."""
    B, H, W, C = x.shape
    _, Ho, Wo, _ = dy.shape
    py, ldy = view(dy, torch.bfloat16)
    px, ldx = view(x, torch.bfloat16)
    L = _L()
    ws = _workspace(L.mi_gdwconv_wgrad_workspace(B, Ho, Wo, C), x.device, "gdw")
    check(L.mi_gdwconv_wgrad(py, ldy, px, ldx, _p(dw), _p(dbias), B, H, W, C, Ho, Wo, stride, pad, int(accumulate), _p(ws), ws.numel(), _stream()), "mi_gdwconv_wgrad")
    if not need_dx:
        return None
    dx = new(B, H, W, C, x.device)
    pd, ldd = view(dx, torch.bfloat16)
    check(L.mi_gdwconv_dgrad(py, ldy, _p(w), pd, ldd, B, H, W, C, Ho, Wo, stride, pad, _stream()), "mi_gdwconv_dgrad")
    return dx


def gcca_fwd(q, k, v):
    """Criss-cross attention core: q, k [B,H,W,Cq], v [B,H,W,C] bf16 views -> (agg [B,H,W,C] bf16, att fp32 [B,H,W,H+W])."""
    B, H, W, Cq = q.shape
    C = v.shape[-1]
    att = torch.empty((B, H, W, H + W), dtype=torch.float32, device=q.device)
    agg = new(B, H, W, C, q.device)
    (pq, ldq), (pk, ldk), (pv, ldv), (po, ldo) = view(q, torch.bfloat16), view(k, torch.bfloat16), view(v, torch.bfloat16), view(agg, torch.bfloat16)
    check(_L().mi_gcca_fwd(pq, ldq, pk, ldk, pv, ldv, _p(att), po, ldo, B, H, W, Cq, C, _stream()), "mi_gcca_fwd")
    return agg, att


def gcca_bwd(q, k, v, att, dagg):
    B, H, W, Cq = q.shape
    C = v.shape[-1]
    dq, dk, dv = new(B, H, W, Cq, q.device), new(B, H, W, Cq, q.device), new(B, H, W, C, q.device)
    de = torch.empty_like(att)
    (pq, ldq), (pk, ldk), (pv, ldv), (pg, ldg) = view(q, torch.bfloat16), view(k, torch.bfloat16), view(v, torch.bfloat16), view(dagg, torch.bfloat16)
    (p1, l1), (p2, l2), (p3, l3) = view(dq), view(dk), view(dv)
    check(_L().mi_gcca_bwd(pq, ldq, pk, ldk, pv, ldv, _p(att), pg, ldg, _p(de), p1, l1, p2, l2, p3, l3, B, H, W, Cq, C, _stream()), "mi_gcca_bwd")
    return dq, dk, dv


def ggate(x, g):
    """x + x * sigmoid(g)"""
    B, H, W, C = x.shape
    out = new(B, H, W, C, x.device)
    (px, ldx), (pg, ldg), (po, ldo) = view(x, torch.bfloat16), view(g, torch.bfloat16), view(out)
    check(_L().mi_ggate(px, ldx, pg, ldg, None, 0, po, ldo, None, 0, B * H * W, C, _stream()), "mi_ggate")
    return out


def ggate_bwd(x, g, dout):
    B, H, W, C = x.shape
    dx, dg = new(B, H, W, C, x.device), new(B, H, W, C, x.device)
    (px, ldx), (pg, ldg), (pd, ldd), (p1, l1), (p2, l2) = view(x, torch.bfloat16), view(g, torch.bfloat16), view(dout, torch.bfloat16), view(dx), view(dg)
    check(_L().mi_ggate(px, ldx, pg, ldg, pd, ldd, p1, l1, p2, l2, B * H * W, C, _stream()), "mi_ggate(bwd)")
    return dx, dg


def gmaxpool(x, k, stride, pad):
    """MaxPool2d(k, stride, pad) on a bf16 NHWC view -> (out, idx uint8 [B,Ho,Wo,C])"""
    B, H, W, C = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = new(B, Ho, Wo, C, x.device)
    idx = torch.empty((B, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    (px, ldx), (po, ldo) = view(x, torch.bfloat16), view(out)
    check(_L().mi_gmaxpool(px, ldx, po, ldo, _p(idx), B, H, W, C, Ho, Wo, k, stride, pad, 0, _stream()), "mi_gmaxpool")
    return out, idx


def gmaxpool_bwd(dout, idx, in_hw, k, stride, pad, dx=None):
    B, Ho, Wo, C = dout.shape
    H, W = in_hw
    if dx is None:
        dx = new(B, H, W, C, dout.device)
    (px, ldx), (po, ldo) = view(dx, torch.bfloat16), view(dout, torch.bfloat16)
    check(_L().mi_gmaxpool(px, ldx, po, ldo, _p(idx), B, H, W, C, Ho, Wo, k, stride, pad, 1, _stream()), "mi_gmaxpool(bwd)")
    return dx


def gce(logits, labels, ignore_index=255, want_grad=True, grad_scale=1.0):
    """CrossEntropyLoss(ignore_index) on NHWC fp32 logits [B,H,W,K] (view), labels int64 [B,H,W] -> (loss_out[4], dlogits or None)."""
    B, H, W, Kc = logits.shape
    pl, ld = view(logits, torch.float32)
    if not (labels.dtype == torch.int64 and labels.is_contiguous() and labels.numel() == B * H * W):
        raise _lib.MiError("gce: labels must be contiguous int64 [B,H,W]")
    L = _L()
    ws = _workspace(L.mi_gce_workspace(B * H * W), logits.device, "gce")
    out = torch.empty(4, dtype=torch.float32, device=logits.device)
    d = torch.empty((B, H, W, Kc), dtype=torch.float32, device=logits.device) if want_grad else None
    check(L.mi_gce(pl, ld, _p(labels), B * H * W, Kc, int(ignore_index), _p(out), _p(d), Kc, ctypes.c_float(grad_scale), _p(ws), ws.numel(), _stream()), "mi_gce")
    return out, d


# ---------------------------------------------------------------------------------------------- the family in fp32 (csrc/gf32.hip): evaluation forward
F32 = torch.float32
ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2
PW_AFFINE, PW_REVERSE, PW_GATE, PW_MULRELU = 0, 1, 2, 3


def _act(relu):
    return ACT_RELU6 if relu == 6 else (ACT_RELU if relu else ACT_NONE)


def gconv_f32(x, w, geom, bias=None, scale=None, shift=None, add=None, relu=False, out=None):
    """act(((conv(x, w) + bias) * scale + shift) + add) in fp32: x [B,H,W,Cin] fp32 view, w the fp32 OIHW master weight."""
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    B, Ha, Wa, Ca = x.shape
    N = w.shape[0]
    Ho, Wo = conv_out_hw(Ha, Wa, *geom)
    if out is None:
        out = new(B, Ho, Wo, N, x.device, F32)
    if tuple(out.shape) != (B, Ho, Wo, N) or tuple(w.shape) != (N, Ca, kh, kw) or not w.is_contiguous() or w.dtype != F32:
        raise _lib.MiError("gconv_f32: out %s / weight %s do not fit input %s, taps %dx%d" % (tuple(out.shape), tuple(w.shape), tuple(x.shape), kh, kw))
    px, ldx = view(x, F32)
    po, ldo = view(out, F32)
    pa, lda = view(add, F32) if add is not None else (None, 0)
    check(_L().mi_gconv_f32(px, ldx, _p(w), _p(bias), _p(scale), _p(shift), pa, lda, _act(relu), po, ldo, B, Ha, Wa, Ca, Ho, Wo, N, kh, kw, sh, sw, ph, pw, dh, dw,
                            _stream()), "mi_gconv_f32")
    return out


def gpool_f32(x, k, stride, pad, mode, out_hw=None, out=None):
    """mode 0 / 1: the two AvgPool2d conventions of Res2Net_v1b.py:40,122 (out_hw given by the caller for the ceil_mode one); 2: MaxPool2d."""
    B, H, W, C = x.shape
    Ho, Wo = out_hw if out_hw is not None else ((H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1)
    if out is None:
        out = new(B, Ho, Wo, C, x.device, F32)
    (px, ldx), (po, ldo) = view(x, F32), view(out, F32)
    check(_L().mi_gpool_f32(px, ldx, po, ldo, B, H, W, C, Ho, Wo, k, stride, pad, mode, _stream()), "mi_gpool_f32")
    return out


def gdwconv_f32(x, w, bias, stride, pad, scale=None, shift=None, relu=False):
    B, H, W, C = x.shape
    Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    out = new(B, Ho, Wo, C, x.device, F32)
    (px, ldx), (po, ldo) = view(x, F32), view(out, F32)
    check(_L().mi_gdwconv_f32(px, ldx, _p(w), _p(bias), _p(scale), _p(shift), _act(relu), po, ldo, B, H, W, C, Ho, Wo, stride, pad, _stream()), "mi_gdwconv_f32")
    return out


def gcca_f32(q, k, v):
    B, H, W, Cq = q.shape
    C = v.shape[-1]
    out = new(B, H, W, C, q.device, F32)
    (pq, ldq), (pk, ldk), (pv, ldv), (po, ldo) = view(q, F32), view(k, F32), view(v, F32), view(out, F32)
    check(_L().mi_gcca_f32(pq, ldq, pk, ldk, pv, ldv, po, ldo, B, H, W, Cq, C, _stream()), "mi_gcca_f32")
    return out


def gpoint_f32(op, a, b=None, scale=None, shift=None, relu=False, out=None):
    B, H, W, C = a.shape
    if out is None:
        out = new(B, H, W, C, a.device, F32)
    (pa, lda), (po, ldo) = view(a, F32), view(out, F32)
    if b is None:
        pb, ldb = None, 0
    elif op == PW_REVERSE:
        if not (b.dtype == F32 and b.is_contiguous() and b.numel() == B * H * W):
            raise _lib.MiError("gpoint_f32: the reverse-attention gate must be contiguous fp32 with one value per pixel")
        pb, ldb = ctypes.c_void_p(b.data_ptr()), 1
    else:
        pb, ldb = view(b, F32)
    check(_L().mi_gpoint_f32(op, pa, lda, pb, ldb, _p(scale), _p(shift), _act(relu), po, ldo, B * H * W, C, _stream()), "mi_gpoint_f32")
    return out
