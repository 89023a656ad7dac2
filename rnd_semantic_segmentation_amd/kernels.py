"""Tensor-level wrappers over the C-ABI (torch is plumbing: device memory + streams).

Layout: every activation handled here is an NHWC-contiguous tensor [B,H,W,C]
(bf16 unless stated).  Callers that hold NCHW-shaped channels_last tensors pass
`x.permute(0, 2, 3, 1)` (a free view).
"""
import ctypes

import torch

from . import _lib
from ._lib import check

EPI_SCALE_BIAS, EPI_RESIDUAL, EPI_RELU, EPI_MASK, EPI_OUT_F32, EPI_ZSPLIT, EPI_WRITE_MASK, EPI_BITMASK, EPI_LEAKY = 1, 2, 4, 8, 16, 32, 64, 128, 256
GATHER_FWD, GATHER_DGRAD = 0, 1
ASPP_ZGW, ASPP_KPAD = 20, 704


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional per-launch timing with HIP events on the launch stream (bench.py): set to a list to collect
# (kernel_name, start_event, stop_event, algorithmic_flops) tuples; None = no overhead.
PROFILE = None


def _timed(name, flops, launch, tag=None):
    if PROFILE is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = launch()
    e1.record()
    PROFILE.append((name, e0, e1, flops, tag))
    return rc


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _chk(t, dtype, name):
    if not t.is_cuda:
        raise _lib.MiError("%s must live on the GPU (no CPU path exists)" % name)
    if t.dtype != dtype:
        raise _lib.MiError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise _lib.MiError("%s must be contiguous" % name)
    return t


def pack_weight_fwd(w, out=None):
    """fp32 OIHW -> bf16 [k*k][O][I]"""
    _chk(w, torch.float32, "w")
    O, I, k, _ = w.shape
    if out is None:
        out = torch.empty((k * k, O, I), dtype=torch.bfloat16, device=w.device)
    check(_lib.lib().mi_pack_weight_fwd(_p(w), _p(out), O, I, k, _stream()), "mi_pack_weight_fwd")
    return out


def pack_weight_dgrad(w, scale=None, out=None):
    """fp32 OIHW -> bf16 [k*k][I][O], FrozenBN scale[o] folded in"""
    _chk(w, torch.float32, "w")
    O, I, k, _ = w.shape
    if out is None:
        out = torch.empty((k * k, I, O), dtype=torch.bfloat16, device=w.device)
    check(_lib.lib().mi_pack_weight_dgrad(_p(w), _p(scale), _p(out), O, I, k, _stream()), "mi_pack_weight_dgrad")
    return out


def pack_weights_multi(wflat, sflat, wp, wpt, table_dev, n_desc, total_blocks):
    """One launch packing every conv weight described by `table_dev` (see include/mi355seg.h)."""
    check(_lib.lib().mi_pack_weights_multi(_p(wflat), _p(sflat), _p(wp), _p(wpt), _p(table_dev), n_desc, total_blocks, _stream()),
          "mi_pack_weights_multi")


def conv_gemm(a, wp, out_hw, ksize=1, stride=1, pad=0, dil=1, mode=GATHER_FWD, scale=None, bias=None, res=None,
              msk=None, relu=False, out_f32=False, zsplit=0, out=None, bits=None, mask_out=None, leaky=0.0, flop_cols=0, wide=None):
    """out[b,ho,wo,n] = epi(sum_{t,c} a[b,src(ho,wo,t),c] * wp[t,n,c]);  a [B,Ha,Wa,Ca] bf16, wp [k*k,N,Ca] bf16.
    msk: bf16 [B,Ho,Wo,N] ReLU mask source; bits: the same mask as packed sign bits (int16 [B,Ho,Wo,N/16]);
    mask_out: int16 [B,Ho,Wo,N/16] receiving the sign bits of the result.
    flop_cols: live columns of a padded ASPP operand (per zsplit plane, or of Ca == 704), for the FLOP accounting only.
    wide: None = the library's own choice of main loop; 8 / 10 = force the wide-tile ping-pong loop (mi_conv_gemm_pp) with that
    many 16-row MFMA tiles per wave (tests, tools/ppexp.py)."""
    _chk(a, torch.bfloat16, "a")
    _chk(wp, torch.bfloat16, "wp")
    B, Ha, Wa, Ca = a.shape
    T, N, Cw = wp.shape
    if T != ksize * ksize or Cw != Ca:
        raise _lib.MiError("packed weight %s does not match ksize=%d, Ca=%d" % (tuple(wp.shape), ksize, Ca))
    Ho, Wo = out_hw
    flags = 0
    if scale is not None:
        flags |= EPI_SCALE_BIAS
        _chk(scale, torch.float32, "scale")
        _chk(bias, torch.float32, "bias")
    if res is not None:
        flags |= EPI_RESIDUAL
        _chk(res, torch.bfloat16, "res")
        assert tuple(res.shape) == (B, Ho, Wo, N)
    if relu:
        flags |= EPI_RELU
    if leaky:
        flags |= EPI_LEAKY          # LeakyReLU(leaky) forward (with relu=True) or its backward (with bits=...)
    if msk is not None:
        flags |= EPI_MASK
        _chk(msk, torch.bfloat16, "msk")
        assert tuple(msk.shape) == (B, Ho, Wo, N)
    if bits is not None:
        flags |= EPI_BITMASK
        _chk(bits, torch.int16, "bits")
        assert tuple(bits.shape) == (B, Ho, Wo, N // 16) and msk is None
        msk = bits
    if mask_out is not None:
        flags |= EPI_WRITE_MASK
        _chk(mask_out, torch.int16, "mask_out")
        assert tuple(mask_out.shape) == (B, Ho, Wo, N // 16)
    if zsplit:
        flags |= EPI_ZSPLIT | EPI_OUT_F32
        if out is None:
            out = torch.empty((N // zsplit, B * Ho * Wo, zsplit), dtype=torch.float32, device=a.device)
    elif out_f32:
        flags |= EPI_OUT_F32
        if out is None:
            out = torch.empty((B, Ho, Wo, N), dtype=torch.float32, device=a.device)
    elif out is None:
        out = torch.empty((B, Ho, Wo, N), dtype=torch.bfloat16, device=a.device)
    _chk(out, torch.float32 if (zsplit or out_f32) else torch.bfloat16, "out")
    if out.numel() != B * Ho * Wo * N:
        raise _lib.MiError("out has %d elements, the conv writes %d" % (out.numel(), B * Ho * Wo * N))
    # algorithmic FLOPs (2 * pixels * C_out * C_in * k^2), padding columns of the ASPP operands excluded
    n_real = N * flop_cols // zsplit if (zsplit and flop_cols) else N
    ca_real = flop_cols if (Ca == ASPP_KPAD and flop_cols) else Ca
    flops = 2.0 * B * Ho * Wo * n_real * ca_real * ksize * ksize
    if wide is not None:
        check(_lib.lib().mi_conv_gemm_pp(_p(a), _p(wp), _p(out), B, Ha, Wa, Ca, Ho, Wo, N, ksize, stride, pad, dil, mode, _p(scale), _p(bias), _p(res),
                                         _p(msk), _p(mask_out), flags, zsplit, float(leaky), int(wide), _stream()), "mi_conv_gemm_pp")
        return out
    kern = "igemm_nt_kernel"
    if PROFILE is not None and _lib.lib().mi_conv_gemm_route(B, Ha, Wa, Ca, Ho, Wo, N, ksize, stride, flags):
        kern = "igemm_pp_kernel"
    check(_timed(kern, flops, lambda: _lib.lib().mi_conv_gemm(
        _p(a), _p(wp), _p(out), B, Ha, Wa, Ca, Ho, Wo, N, ksize, stride, pad, dil, mode,
        _p(scale), _p(bias), _p(res), _p(msk), _p(mask_out), flags, zsplit, float(leaky), _stream()),
        tag=("dgrad" if mode == GATHER_DGRAD else "fwd", ksize, Ca, N, B * Ho * Wo, flags, dil)), "mi_conv_gemm")
    return out


_ws_cache = {}


def conv_gemm_stats(a, wp, out_hw, ksize, stride, pad, dil, pilot, bn=None):
    """conv_gemm with a plain bf16 store that also returns the BatchNorm statistics of its output, taken in the epilogue: (out, sums, fin) with
    sums[0][n] = sum (out - pilot[n]), sums[1][n] = sum (out - pilot[n])^2 ([2, N] fp32) - bn_colsum2(out, pilot) without the extra read.
    bn (an nn.BatchNorm2d whose statistics are NOT shared across ranks): the last reduction launch also finalizes them (bn_finalize's result `fin`
    and running-statistics update, count = the pixels of this tensor); otherwise fin is None."""
    _chk(a, torch.bfloat16, "a")
    _chk(wp, torch.bfloat16, "wp")
    _chk(pilot, torch.float32, "pilot")
    B, Ha, Wa, Ca = a.shape
    T, N, Cw = wp.shape
    if T != ksize * ksize or Cw != Ca:
        raise _lib.MiError("packed weight %s does not match ksize=%d, Ca=%d" % (tuple(wp.shape), ksize, Ca))
    Ho, Wo = out_hw
    out = torch.empty((B, Ho, Wo, N), dtype=torch.bfloat16, device=a.device)
    sums = torch.empty((2, N), dtype=torch.float32, device=a.device)
    ws = _workspace(int(_lib.lib().mi_conv_gemm_stats_workspace(B * Ho * Wo, N)), a.device, "conv_stats")
    fin, fa = None, (None, None, None, None, None, 0.0, 0.0, None)
    if bn is not None:
        track = bn.track_running_stats and bn.running_mean is not None
        if track and bn.momentum is None:
            raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative average) is not used by the reference")
        fin = torch.empty((4, N), dtype=torch.float32, device=a.device)
        fa = (_p(bn.weight.detach()), _p(bn.bias.detach()), _p(bn.running_mean) if track else None, _p(bn.running_var) if track else None,
              _p(bn.num_batches_tracked) if track else None, float(bn.momentum or 0.0), float(bn.eps), _p(fin))
    flops = 2.0 * B * Ho * Wo * N * Ca * ksize * ksize
    _timed("igemm_stats", flops, lambda: check(_lib.lib().mi_conv_gemm_stats(_p(a), _p(wp), _p(out), B, Ha, Wa, Ca, Ho, Wo, N, ksize, stride, pad, dil, _p(pilot),
                                                                             _p(sums), _p(ws), ws.numel(), *fa, _stream()), "mi_conv_gemm_stats"),
           ("fwd", ksize, Ca, N, B * Ho * Wo, 512, dil))
    return out, sums, fin


def _workspace(nbytes, device, tag="ws"):
    # one buffer per (purpose, device, stream): launches on one stream are ordered, so they can share it; two streams cannot
    key = (tag, str(device), torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def conv_chain(a, w_first, res, w_second, scale1=None, shift1=None, scale2=None, shift2=None, bits1=None, bits2=None, mid=None, out=None,
               bits1_out=None, bits2_out=None, grid=0):
    """Two chained 1x1 convs in one launch (csrc/chain.hip).  Forward (scale1 given): mid = relu(scale1 * a.w_first^T + shift1 + res),
    out = relu(scale2 * mid.w_second^T + shift2), returns (mid, out, bits(mid), bits(out)).  Backward (bits1 given): mid = (a.w_first^T + res) masked by
    bits1, out = (mid.w_second^T) masked by bits2, returns (mid, out).  a [B,H,W,K1], res [B,H,W,N1] bf16 NHWC; w_first [1,N1,K1], w_second [1,N2,N1]."""
    _chk(a, torch.bfloat16, "a")
    _chk(res, torch.bfloat16, "res")
    _chk(w_first, torch.bfloat16, "w_first")
    _chk(w_second, torch.bfloat16, "w_second")
    B, H, W, K1 = a.shape
    N1, N2 = w_first.shape[-2], w_second.shape[-2]
    if w_first.shape[-1] != K1 or w_second.shape[-1] != N1 or tuple(res.shape) != (B, H, W, N1):
        raise _lib.MiError("conv_chain: a %s, w_first %s, res %s, w_second %s do not chain" % (tuple(a.shape), tuple(w_first.shape), tuple(res.shape), tuple(w_second.shape)))
    backward = bits1 is not None
    M = B * H * W
    if mid is None:
        mid = torch.empty((B, H, W, N1), dtype=torch.bfloat16, device=a.device)
    if out is None:
        out = torch.empty((B, H, W, N2), dtype=torch.bfloat16, device=a.device)
    _chk(mid, torch.bfloat16, "mid")
    _chk(out, torch.bfloat16, "out")
    if mid.numel() != M * N1 or out.numel() != M * N2:
        raise _lib.MiError("conv_chain: mid / out have the wrong size")
    if backward:
        _chk(bits1, torch.int16, "bits1")
        _chk(bits2, torch.int16, "bits2")
        assert bits1.numel() == M * N1 // 16 and bits2.numel() == M * N2 // 16
    else:
        for t, n in ((scale1, N1), (shift1, N1), (scale2, N2), (shift2, N2)):
            _chk(t, torch.float32, "scale / shift")
            assert t.numel() == n
        if bits1_out is None:
            bits1_out = torch.empty((B, H, W, N1 // 16), dtype=torch.int16, device=a.device)
        if bits2_out is None:
            bits2_out = torch.empty((B, H, W, N2 // 16), dtype=torch.int16, device=a.device)
        _chk(bits1_out, torch.int16, "bits1_out")
        _chk(bits2_out, torch.int16, "bits2_out")
        assert bits1_out.numel() == M * N1 // 16 and bits2_out.numel() == M * N2 // 16
    flops = 2.0 * M * N1 * (K1 + N2)
    check(_timed("chain_kernel", flops, lambda: _lib.lib().mi_conv_chain(
        _p(a), _p(w_first), _p(res), _p(mid), _p(w_second), _p(out), M, K1, N1, N2, _p(scale1), _p(shift1), _p(scale2), _p(shift2),
        _p(bits1), _p(bits2), _p(bits1_out), _p(bits2_out), 1 if backward else 0, int(grid), _stream()),
        tag=("chain_bwd" if backward else "chain_fwd", 1, K1, N1, M, 0, 1)), "mi_conv_chain")
    return (mid, out) if backward else (mid, out, bits1_out, bits2_out)


class WgradBatch:
    """The slab reducers of up to eight weight gradients as ONE launch (mi_conv_wgrad_partial + mi_conv_wgrad_reduce): conv_wgrad(..., batch=b) runs the
    main kernel only and keeps its split-K slabs in a workspace of its own; b.flush() (same stream, after the last of them) sums them all.  Same bits as
    the one-call form.  A full batch flushes itself."""
    MAX = 8

    def __init__(self):
        self._n = 0
        self._size = int(_lib.lib().mi_conv_wgrad_job_bytes())
        self._jobs = ctypes.create_string_buffer(self.MAX * self._size)
        self._keep = []

    def slot(self):
        if self._n == self.MAX:
            self.flush()
        return self._n, ctypes.c_void_p(ctypes.addressof(self._jobs) + self._n * self._size)

    def added(self, *tensors):
        self._n += 1
        self._keep.extend(tensors)

    def flush(self):
        if self._n:
            check(_timed("wgrad_reduce_multi_kernel", 0.0, lambda: _lib.lib().mi_conv_wgrad_reduce(self._jobs, self._n, _stream()), tag=("wgrad_reduce", 1, 0, 0, 0, 0, 1)),
                  "mi_conv_wgrad_reduce")
            self._n, self._keep = 0, []


def conv_wgrad(dy, x, dw, ksize=1, stride=1, pad=0, dil=1, scale=None, accumulate=False, out_map=0, ncls=0, batch=None):
    """dw[o,i,ky,kx] (+)= scale[o] * sum_m dy[m,o] x[src(m,t),i];  dy [B,Ho,Wo,O], x [B,Ha,Wa,I] bf16; dw fp32.
    out_map=1 (ASPP): dy is the im2col matrix with 36*ncls live columns, dw the 4 stacked [ncls,I,3,3] gradients.
    batch (a WgradBatch): the slab reducer is deferred to batch.flush()."""
    _chk(dy, torch.bfloat16, "dy")
    _chk(x, torch.bfloat16, "x")
    _chk(dw, torch.float32, "dw")
    B, Ho, Wo, O = dy.shape
    _, Ha, Wa, I = x.shape
    L = _lib.lib()
    need = L.mi_conv_wgrad_workspace(B, Ho, Wo, O, I, ksize)
    ws = _workspace(need, dy.device, "wgrad")
    if out_map == 1 and not 0 < 36 * ncls <= O:
        raise _lib.MiError("conv_wgrad(out_map=1) needs ncls with 36*ncls <= %d, got %r" % (O, ncls))
    o_real = 36 * ncls if out_map == 1 else O
    flops = 2.0 * B * Ho * Wo * o_real * I * ksize * ksize
    kern = "wgrad_tn_kernel+reduce"
    if PROFILE is not None:
        kern = ("wgrad_tn_kernel", "wgrad_tn256_kernel", "wgrad_p3_kernel", "wgrad_q3_kernel", "wgrad_s4_kernel")[
            L.mi_conv_wgrad_route(B, Ha, Wa, I, Ho, Wo, O, ksize, stride, pad, dil, out_map)] + "+reduce"
    if batch is not None:
        idx, job = batch.slot()
        ws = _workspace(need, dy.device, "wgrad_b%d" % idx)             # its slabs must survive until the flush: one buffer per batch slot and stream
        check(_timed(kern.replace("+reduce", ""), flops, lambda: L.mi_conv_wgrad_partial(
            _p(dy), _p(x), _p(dw), B, Ha, Wa, I, Ho, Wo, O, ksize, stride, pad, dil, _p(scale),
            int(accumulate), out_map, int(ncls), dw.numel(), _p(ws), ws.numel(), job, _stream()), tag=("wgrad", ksize, I, O, B * Ho * Wo, out_map, dil)), "mi_conv_wgrad_partial")
        batch.added(ws, dw, scale)
        return dw
    check(_timed(kern, flops, lambda: L.mi_conv_wgrad(
        _p(dy), _p(x), _p(dw), B, Ha, Wa, I, Ho, Wo, O, ksize, stride, pad, dil, _p(scale),
        int(accumulate), out_map, int(ncls), dw.numel(), _p(ws), ws.numel(), _stream()), tag=("wgrad", ksize, I, O, B * Ho * Wo, out_map, dil)), "mi_conv_wgrad")
    return dw


def _rates(rates):
    return (ctypes.c_int * 4)(*[int(r) for r in rates])


def aspp_pack_fwd(w4, out=None):
    """[4,K,C,3,3] fp32 -> Wall bf16 [1, 720, C] (packed-weight form for conv_gemm, ksize 1)"""
    _chk(w4, torch.float32, "w4")
    R, K, C = w4.shape[:3]
    if out is None:
        out = torch.empty((1, 36 * ASPP_ZGW, C), dtype=torch.bfloat16, device=w4.device)
    check(_lib.lib().mi_aspp_pack_fwd(_p(w4), _p(out), C, K, _stream()), "mi_aspp_pack_fwd")
    return out


def aspp_pack_dgrad(w4, out=None):
    """[4,K,C,3,3] fp32 -> WallT bf16 [1, C, 704]"""
    _chk(w4, torch.float32, "w4")
    R, K, C = w4.shape[:3]
    if out is None:
        out = torch.empty((1, C, ASPP_KPAD), dtype=torch.bfloat16, device=w4.device)
    check(_lib.lib().mi_aspp_pack_dgrad(_p(w4), _p(out), C, K, _stream()), "mi_aspp_pack_dgrad")
    return out


def aspp_col2im(z, bias4, B, H, W, K, rates):
    _chk(z, torch.float32, "z")
    _chk(bias4, torch.float32, "bias4")
    low = torch.empty((B, H, W, K), dtype=torch.float32, device=z.device)
    check(_timed("aspp_col2im_kernel", 0.0, lambda: _lib.lib().mi_aspp_col2im(_p(z), _p(bias4), _p(low), B, H, W, K, _rates(rates), _stream()),
                 tag=("aspp_aux", 0, 0, 0, B * H * W, 0, 0)), "mi_aspp_col2im")
    return low


def aspp_im2col(dlow, rates):
    _chk(dlow, torch.float32, "dlow")
    B, H, W, K = dlow.shape
    g = torch.empty((B, H, W, ASPP_KPAD), dtype=torch.bfloat16, device=dlow.device)
    check(_timed("aspp_im2col_kernel", 0.0, lambda: _lib.lib().mi_aspp_im2col(_p(dlow), _p(g), B, H, W, K, _rates(rates), _stream()),
                 tag=("aspp_aux", 0, 0, 0, B * H * W, 0, 0)), "mi_aspp_im2col")
    return g


def aspp_bias_grad(dlow, dbias4, accumulate=False):
    _chk(dlow, torch.float32, "dlow")
    _chk(dbias4, torch.float32, "dbias4")
    B, H, W, K = dlow.shape
    L = _lib.lib()
    ws = _workspace(L.mi_colsum_workspace(B * H * W, K), dlow.device, "colsum")
    check(_timed("colsum_kernels", 0.0, lambda: L.mi_aspp_bias_grad(_p(dlow), _p(dbias4), B * H * W, K, int(accumulate), _p(ws), ws.numel(), _stream()),
                 tag=("aspp_aux", 0, 0, 0, B * H * W, 0, 0)), "mi_aspp_bias_grad")
    return dbias4


def upsample_ac_fwd(low, size):
    """low [B,h,w,K] fp32 NHWC -> up [B,K,H,W] fp32 NCHW"""
    _chk(low, torch.float32, "low")
    B, h, w, K = low.shape
    H, W = size
    up = torch.empty((B, K, H, W), dtype=torch.float32, device=low.device)
    check(_lib.lib().mi_upsample_ac_fwd(_p(low), _p(up), B, h, w, K, H, W, _stream()), "mi_upsample_ac_fwd")
    return up


def upsample_ac_bwd(dup, low_hw):
    _chk(dup, torch.float32, "dup")
    B, K, H, W = dup.shape
    h, w = low_hw
    dlow = torch.empty((B, h, w, K), dtype=torch.float32, device=dup.device)
    check(_lib.lib().mi_upsample_ac_bwd(_p(dup), _p(dlow), B, h, w, K, H, W, _stream()), "mi_upsample_ac_bwd")
    return dlow


def softmax_ce_fwd(logits, labels, ignore_index=255):
    """logits [B,K,H,W] fp32, labels [B,H,W] int64 -> loss_out fp32[4] = (mean loss, n_valid, out-of-range labels, scratch)"""
    _chk(logits, torch.float32, "logits")
    _chk(labels, torch.int64, "labels")
    B, K, H, W = logits.shape
    L = _lib.lib()
    ws = _workspace(L.mi_ce_workspace(B, H, W), logits.device, "ce")
    out = torch.empty(4, dtype=torch.float32, device=logits.device)
    check(L.mi_softmax_ce_fwd(_p(logits), _p(labels), _p(out), B, K, H, W, ignore_index, _p(ws), ws.numel(), _stream()),
          "mi_softmax_ce_fwd")
    return out


def softmax_ce_bwd(logits, labels, loss_out, grad_scale=1.0, ignore_index=255):
    _chk(logits, torch.float32, "logits")
    _chk(labels, torch.int64, "labels")
    B, K, H, W = logits.shape
    d = torch.empty_like(logits)
    check(_lib.lib().mi_softmax_ce_bwd(_p(logits), _p(labels), _p(loss_out), _p(d), B, K, H, W, ignore_index,
                                        float(grad_scale), _stream()), "mi_softmax_ce_bwd")
    return d


def upsample_ce(low, labels, want_grad=True, grad_scale=1.0, ignore_index=255, align_corners=True):
    """Fused classifier-upsample + CrossEntropyLoss.  Returns (loss_out[4], dlow or None).  align_corners False: F.interpolate(size=) default."""
    _chk(low, torch.float32, "low")
    _chk(labels, torch.int64, "labels")
    B, h, w, K = low.shape
    _, H, W = labels.shape
    L = _lib.lib()
    ws = _workspace(L.mi_upsample_ce_workspace(B, h, w, K, H, W), low.device, "upce")
    out = torch.empty(4, dtype=torch.float32, device=low.device)
    dlow = torch.empty_like(low) if want_grad else None
    check(L.mi_upsample_ce_ex(_p(low), _p(labels), _p(out), _p(dlow), B, h, w, K, H, W, ignore_index, float(grad_scale), int(bool(align_corners)),
                              _p(ws), ws.numel(), _stream()), "mi_upsample_ce")
    return out, dlow


def check_labels(loss_out, num_classes, what="labels"):
    """Raise if the CE kernels saw labels outside [0, num_classes) that are not ignore_index (torch's device assert).
    Synchronises (reads one float): call it where the loss value is fetched anyway."""
    bad = int(loss_out[2].item())
    if bad:
        raise ValueError("%s: %d label values lie outside [0, %d) and are not ignore_index - torch.nn.CrossEntropyLoss would "
                         "raise a device assert; map the label ids to train ids first" % (what, bad, num_classes))


def upsample_softmax(low, size, want_pred=True):
    _chk(low, torch.float32, "low")
    B, h, w, K = low.shape
    H, W = size
    probs = torch.empty((B, K, H, W), dtype=torch.float32, device=low.device)
    pred = torch.empty((B, H, W), dtype=torch.uint8, device=low.device) if want_pred else None
    check(_lib.lib().mi_upsample_softmax(_p(low), _p(probs), _p(pred), B, h, w, K, H, W, _stream()), "mi_upsample_softmax")
    return probs, pred


def stem_pool_fwd(y, scale, shift):
    """y [B,Hc,Wc,C] bf16 (conv1 output) -> (pool [B,Hp,Wp,C] bf16, idx uint8): FrozenBN + ReLU + maxpool 3x3/2/1."""
    _chk(y, torch.bfloat16, "y")
    B, Hc, Wc, C = y.shape
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    pool = torch.empty((B, Hp, Wp, C), dtype=torch.bfloat16, device=y.device)
    idx = torch.empty((B, Hp, Wp, C), dtype=torch.uint8, device=y.device)
    check(_lib.lib().mi_stem_pool_fwd(_p(y), _p(scale), _p(shift), _p(pool), _p(idx), B, Hc, Wc, C, Hp, Wp, _stream()), "mi_stem_pool_fwd")
    return pool, idx


def stem_pool_bwd(dpool, idx, scale, conv_hw):
    _chk(dpool, torch.bfloat16, "dpool")
    _chk(idx, torch.uint8, "idx")
    B, Hp, Wp, C = dpool.shape
    Hc, Wc = conv_hw
    dy = torch.empty((B, Hc, Wc, C), dtype=torch.bfloat16, device=dpool.device)
    check(_lib.lib().mi_stem_pool_bwd(_p(dpool), _p(idx), _p(scale), _p(dy), B, Hc, Wc, C, Hp, Wp, _stream()), "mi_stem_pool_bwd")
    return dy


def stem_im2col(x, ncols):
    """Patch matrix of the 7x7 / stride 2 / pad 3 stem conv: x [B,3,H,W] bf16 channels_last -> [B,Ho,Wo,ncols] bf16 (147 live columns)."""
    _chk_dtype = x.dtype
    if _chk_dtype != torch.bfloat16 or not x.is_cuda:
        raise _lib.MiError("stem_im2col: x must be a bf16 GPU tensor")
    B, _, H, W = x.shape
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    xn = x.permute(0, 2, 3, 1)
    if not xn.is_contiguous():
        xn = xn.contiguous()
    col = torch.empty((B, Ho, Wo, ncols), dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().mi_stem_im2col(_p(xn), _p(col), B, H, W, Ho, Wo, ncols, _stream()), "mi_stem_im2col")
    return col


def stem_conv_fwd(x, weight):
    """The stem conv as patch matrix + plain GEMM on the implicit-GEMM kernel: returns (y [B,Ho,Wo,64] bf16 NHWC, the patch matrix, which the
    weight gradient reuses)."""
    col = stem_im2col(x, 192)
    O = weight.shape[0]
    wp = torch.zeros((1, O, 192), dtype=torch.bfloat16, device=x.device)
    wp[0, :, :147] = weight.detach().reshape(O, 147).to(torch.bfloat16)
    return conv_gemm(col, wp, (col.shape[1], col.shape[2]), 1, 1, 0, 1, GATHER_FWD), col


def stem_wgrad(dy, x=None, col=None):
    """Weight gradient of the stem conv, deterministic: dy [B,Ho,Wo,64] bf16 NHWC and either the forward's patch matrix `col` or the input
    x [B,3,H,W] bf16 channels_last (a 160-column matrix is built then) -> fp32 [64,3,7,7], by the 1x1 weight-gradient kernel with its
    fixed-order slab reduction."""
    _chk(dy, torch.bfloat16, "dy")
    if col is None:
        col = stem_im2col(x, 160)
    O, n = dy.shape[-1], col.shape[-1]
    dw = torch.empty((O, n, 1, 1), dtype=torch.float32, device=dy.device)
    conv_wgrad(dy, col, dw, 1, 1, 0, 1)
    return dw.view(O, n)[:, :147].reshape(O, 3, 7, 7)


def bias_grad_bf16(dy, db, accumulate=False):
    """db[n] (+)= sum over pixels of dy[..., n];  dy bf16 [B,H,W,N]"""
    _chk(dy, torch.bfloat16, "dy")
    _chk(db, torch.float32, "db")
    N = dy.shape[-1]
    L = _lib.lib()
    ws = _workspace(L.mi_colsum_workspace(dy.numel() // N, N), dy.device, "colsum")
    check(L.mi_bias_grad_bf16(_p(dy), _p(db), dy.numel() // N, N, int(accumulate), _p(ws), ws.numel(), _stream()), "mi_bias_grad_bf16")
    return db


def upsample_softce(seg_low, d_low, size, domain, temperature=1.8, clip=0.9, want_grad=True, grad_scale=1.0):
    """Fused soft-label CE of the FADA step.  seg_low [B,h,w,K] fp32 (detached), d_low [B,h,w,ldD] fp32 (first 2K used).
    Returns (loss_out[2], dd_low or None)."""
    _chk(seg_low, torch.float32, "seg_low")
    _chk(d_low, torch.float32, "d_low")
    B, h, w, Kc = seg_low.shape
    ldD = d_low.shape[-1]
    H, W = size
    L = _lib.lib()
    ws = _workspace(L.mi_upsample_softce_workspace(B, h, w, Kc, H, W), seg_low.device, "softce")
    out = torch.empty(2, dtype=torch.float32, device=seg_low.device)
    dd = torch.empty_like(d_low) if want_grad else None
    check(L.mi_upsample_softce(_p(seg_low), 1.0 / float(temperature), float(clip), _p(d_low), ldD, int(domain), _p(out), _p(dd), B, h, w, Kc,
                               H, W, float(grad_scale), _p(ws), ws.numel(), _stream()), "mi_upsample_softce")
    return out, dd


def adam_step(p, g, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_clamp=None):
    for t, n in ((p, "p"), (g, "g"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, torch.float32, n)
    if grad_clamp is not None:          # clip_gradient(optimizer, c) + Adam.step (pranet_trainer.py:59-60): g is clamped in place
        check(_lib.lib().mi_adam_step_clamped(_p(p), _p(g), _p(exp_avg), _p(exp_avg_sq), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                                              int(step), float(grad_clamp), _stream()), "mi_adam_step_clamped")
        return
    check(_lib.lib().mi_adam_step(_p(p), _p(g), _p(exp_avg), _p(exp_avg_sq), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                                  int(step), _stream()), "mi_adam_step")


def adam_step_dev(p, g, exp_avg, exp_avg_sq, hyper):
    """adam_step with (lr, beta1, beta2, eps, grad_clamp or <= 0, step) in a 6-float device tensor (graph-capturable: see mi355seg.h)."""
    for t, n in ((p, "p"), (g, "g"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"), (hyper, "hyper")):
        _chk(t, torch.float32, n)
    check(_lib.lib().mi_adam_step_dev(_p(p), _p(g), _p(exp_avg), _p(exp_avg_sq), p.numel(), _p(hyper), _stream()), "mi_adam_step_dev")


def sgd_step(p, g, buf, lr, momentum, weight_decay):
    for t, n in ((p, "p"), (g, "g"), (buf, "buf")):
        _chk(t, torch.float32, n)
    check(_lib.lib().mi_sgd_step(_p(p), _p(g), _p(buf), p.numel(), float(lr), float(momentum), float(weight_decay), _stream()),
          "mi_sgd_step")


def sgd_step_dev(p, g, buf, hyper):
    """sgd_step with (lr, momentum, weight_decay) in a 3-float device tensor (graph-capturable: see mi355seg.h)."""
    for t, n in ((p, "p"), (g, "g"), (buf, "buf"), (hyper, "hyper")):
        _chk(t, torch.float32, n)
    check(_lib.lib().mi_sgd_step_dev(_p(p), _p(g), _p(buf), p.numel(), _p(hyper), _stream()), "mi_sgd_step_dev")


def relu_mask(x, msk, out=None):
    """y = msk > 0 ? x : 0;  msk is a bf16 tensor of x's shape, or packed sign bits (int16, x.numel()/16 words)."""
    _chk(x, torch.bfloat16, "x")
    bits = msk.dtype == torch.int16
    _chk(msk, torch.int16 if bits else torch.bfloat16, "msk")
    if out is None:
        out = torch.empty_like(x)
    check(_lib.lib().mi_relu_mask(_p(x), _p(msk), _p(out), x.numel(), int(bits), _stream()), "mi_relu_mask")
    return out


def frozen_bn_fold(w, b, mean, var):
    for t in (w, b, mean, var):
        _chk(t, torch.float32, "bn buffer")
    scale, shift = torch.empty_like(w), torch.empty_like(w)
    check(_lib.lib().mi_frozen_bn_fold(_p(w), _p(b), _p(mean), _p(var), _p(scale), _p(shift), w.numel(), _stream()),
          "mi_frozen_bn_fold")
    return scale, shift


# ---------------------------------------------------------------------------------------------- PraNet structure loss
def structure_loss(pred, mask, want_grad=True, grad_scale=1.0):
    """pranet_trainer.py:22-31 on fp32 maps [B,1,H,W] (or [B,H,W]): returns (loss scalar tensor, d loss / d pred * grad_scale or None)."""
    _chk(pred, torch.float32, "pred")
    _chk(mask, torch.float32, "mask")
    if pred.shape != mask.shape or (pred.dim() == 4 and pred.shape[1] != 1):
        raise _lib.MiError("structure_loss: pred / mask must be [B,1,H,W] of the same shape, got %s / %s" % (tuple(pred.shape), tuple(mask.shape)))
    B, H, W = pred.shape[0], pred.shape[-2], pred.shape[-1]
    L = _lib.lib()
    ws = torch.empty(int(L.mi_structure_loss_workspace(B, H, W)), dtype=torch.uint8, device=pred.device)
    out = torch.empty(1 + 2 * B, dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    check(L.mi_structure_loss(_p(pred), _p(mask), B, H, W, _p(out), _p(grad), ctypes.c_float(grad_scale), _p(ws), ws.numel(), _stream()),
          "mi_structure_loss")
    return out[0], grad


# ---------------------------------------------------------------------------------------------- trainable BatchNorm2d (NHWC bf16)
_bn_ws = {}


def _bn_workspace(dev, M, C):
    need = int(_lib.lib().mi_bn_workspace(M, C))
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    w = _bn_ws.get(key)
    if w is None or w.numel() < need:
        w = _bn_ws[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    return w


def bn_colsum(y, mean=None):
    """Raw per-channel sums over the pixels of a bf16 NHWC tensor: sum y, or sum (y - mean)^2 when `mean` is given."""
    _chk(y, torch.bfloat16, "y")
    C = y.shape[-1]
    M = y.numel() // C
    out = torch.empty(C, dtype=torch.float32, device=y.device)
    ws = _bn_workspace(y.device, M, C)
    _timed("bn_kernels", 0.0, lambda: check(_lib.lib().mi_bn_colsum(_p(y), _p(mean), M, C, _p(out), _p(ws), ws.numel(), _stream()), "mi_bn_colsum"),
           ("bn", 0, C, C, M, 0, 0))
    return out


def bn_colsum2(y, pilot):
    """One pass: raw (sum (y - pilot), sum (y - pilot)^2) per channel - two rows of one [2, C] buffer (one all-reduce when synchronised)."""
    _chk(y, torch.bfloat16, "y")
    C = y.shape[-1]
    M = y.numel() // C
    s1, s2 = torch.empty((2, C), dtype=torch.float32, device=y.device)
    ws = _bn_workspace(y.device, M, C)
    _timed("bn_kernels", 0.0, lambda: check(_lib.lib().mi_bn_colsum2(_p(y), _p(pilot), M, C, _p(s1), _p(s2), _p(ws), ws.numel(), _stream()), "mi_bn_colsum2"),
           ("bn", 4, C, C, M, 0, 0))
    return s1, s2


def bn_finalize(s1, s2, pilot, count, bn):
    """[mean | invstd | gamma * invstd | beta - mean * gamma * invstd] ([4, C] fp32) from the pilot-form sums over `count` pixels, and torch's
    running-statistics update of `bn` (an nn.BatchNorm2d), in one launch."""
    C = s1.numel()
    out = torch.empty((4, C), dtype=torch.float32, device=s1.device)
    track = bn.track_running_stats and bn.running_mean is not None
    if track and bn.momentum is None:
        raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative average) is not used by the reference")
    check(_lib.lib().mi_bn_finalize(_p(s1), _p(s2), _p(pilot), float(count), _p(bn.weight.detach()), _p(bn.bias.detach()),
                                    _p(bn.running_mean) if track else None, _p(bn.running_var) if track else None,
                                    _p(bn.num_batches_tracked) if track else None, float(bn.momentum or 0.0), float(bn.eps), _p(out), C, _stream()),
          "mi_bn_finalize")
    return out


def bn_apply(y, mean, scale, beta, res=None, relu=False, want_mask=False):
    """relu?((y - mean) * scale + beta (+ res)) as bf16 NHWC (+ packed sign bits)."""
    _chk(y, torch.bfloat16, "y")
    C = y.shape[-1]
    M = y.numel() // C
    out = torch.empty_like(y)
    bits = torch.empty(y.shape[:-1] + (C // 16,), dtype=torch.int16, device=y.device) if want_mask else None
    _timed("bn_kernels", 0.0, lambda: check(_lib.lib().mi_bn_apply(_p(y), _p(mean), _p(scale), _p(beta), _p(res), _p(out), _p(bits), int(relu), M, C,
                                                                   _stream()), "mi_bn_apply"), ("bn", 1, C, C, M, 0, 0))
    return (out, bits) if want_mask else out


def bn_bwd_colsums(g, y, mean, invstd, relu_bits=None, out=None):
    """Raw (sum g, sum g * xhat) per channel; with relu_bits g counts only where the layer's output was positive.  out: (dbeta, dgamma) to
    write into (the parameters' gradient slots)."""
    _chk(g, torch.bfloat16, "g")
    _chk(y, torch.bfloat16, "y")
    C = y.shape[-1]
    M = y.numel() // C
    if out is None:
        dbeta, dgamma = torch.empty((2, C), dtype=torch.float32, device=y.device)
    else:
        dbeta, dgamma = out
        _chk(dbeta, torch.float32, "dbeta")
        _chk(dgamma, torch.float32, "dgamma")
    ws = _bn_workspace(y.device, M, C)
    _timed("bn_kernels", 0.0, lambda: check(_lib.lib().mi_bn_bwd_colsums(_p(g), _p(y), _p(mean), _p(invstd), _p(relu_bits), M, C, _p(dbeta), _p(dgamma),
                                                                         _p(ws), ws.numel(), _stream()), "mi_bn_bwd_colsums"), ("bn", 2, C, C, M, 0, 0))
    return dbeta, dgamma


def bn_bwd_apply(g, y, mean, invstd, gamma, dbeta, dgamma, count, relu_bits=None):
    """dy of BatchNorm given the (possibly all-reduced) raw sums and the pixel count they were taken over."""
    C = y.shape[-1]
    M = y.numel() // C
    dy = torch.empty_like(y)
    _timed("bn_kernels", 0.0, lambda: check(_lib.lib().mi_bn_bwd_apply(_p(g), _p(y), _p(mean), _p(invstd), _p(gamma), _p(dbeta), _p(dgamma),
                                                                       ctypes.c_float(1.0 / count), _p(relu_bits), _p(dy), M, C, _stream()), "mi_bn_bwd_apply"),
           ("bn", 3, C, C, M, 0, 0))
    return dy


# ---------------------------------------------------------------------------------------------- exact-fp32 evaluation path
def pack_weight_f32(w, out=None):
    """fp32 OIHW -> fp32 [k*k][O][I] (a free view for 1x1)."""
    _chk(w, torch.float32, "w")
    O, I, k, _ = w.shape
    if k == 1:
        return w.detach().view(1, O, I)
    if out is None:
        out = torch.empty((k * k, O, I), dtype=torch.float32, device=w.device)
    check(_lib.lib().mi_pack_weight_f32(_p(w), _p(out), O, I, k, _stream()), "mi_pack_weight_f32")
    return out


def conv_f32(a, wp, out_hw, ksize=1, stride=1, pad=0, dil=1, scale=None, bias=None, res=None, relu=False, out=None):
    """fp32 implicit-GEMM conv on the f32 MFMA: a [B,Ha,Wa,Ca] fp32 NHWC, wp [k*k,N,Ca] fp32 -> [B,Ho,Wo,N] fp32."""
    _chk(a, torch.float32, "a")
    _chk(wp, torch.float32, "wp")
    B, Ha, Wa, Ca = a.shape
    T, N, Cw = wp.shape
    if T != ksize * ksize or Cw != Ca:
        raise _lib.MiError("packed weight %s does not match ksize=%d, Ca=%d" % (tuple(wp.shape), ksize, Ca))
    Ho, Wo = out_hw
    flags = 0
    if bias is not None:
        flags |= EPI_SCALE_BIAS
        _chk(bias, torch.float32, "bias")
        if scale is not None:
            _chk(scale, torch.float32, "scale")
    if res is not None:
        flags |= EPI_RESIDUAL
        _chk(res, torch.float32, "res")
        assert tuple(res.shape) == (B, Ho, Wo, N)
    if relu:
        flags |= EPI_RELU
    if out is None:
        out = torch.empty((B, Ho, Wo, N), dtype=torch.float32, device=a.device)
    _chk(out, torch.float32, "out")
    flops = 2.0 * B * Ho * Wo * N * Ca * ksize * ksize
    check(_timed("igemm_f32_kernel", flops, lambda: _lib.lib().mi_conv_f32(
        _p(a), _p(wp), _p(out), B, Ha, Wa, Ca, Ho, Wo, N, ksize, stride, pad, dil, _p(scale), _p(bias), _p(res), flags, _stream()),
        tag=("f32", ksize, Ca, N, B * Ho * Wo, flags)), "mi_conv_f32")
    return out


def stem_f32(x, w, scale, shift):
    """x [B,3,H,W] fp32 NCHW -> relu(bn(conv7x7/2)) [B,Hc,Wc,64] fp32 NHWC."""
    _chk(x, torch.float32, "x")
    _chk(w, torch.float32, "w")
    B, C, H, W = x.shape
    if C != 3 or tuple(w.shape) != (64, 3, 7, 7):
        raise _lib.MiError("stem_f32 is the 3 -> 64 channel 7x7 stem (got x %s, w %s)" % (tuple(x.shape), tuple(w.shape)))
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, Hc, Wc, 64), dtype=torch.float32, device=x.device)
    check(_lib.lib().mi_stem_f32(_p(x), _p(w), _p(scale), _p(shift), _p(y), B, H, W, _stream()), "mi_stem_f32")
    return y


def maxpool_f32(y):
    _chk(y, torch.float32, "y")
    B, Hc, Wc, C = y.shape
    pool = torch.empty((B, (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1, C), dtype=torch.float32, device=y.device)
    check(_lib.lib().mi_maxpool_f32(_p(y), _p(pool), B, Hc, Wc, C, _stream()), "mi_maxpool_f32")
    return pool
