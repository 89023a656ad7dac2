"""ORACLE (test infrastructure, never shipped as the product path).

Plain numpy restatements of every arithmetic step on the DeepLabV2-R101 + ASPP
hot path of taintpro98/rnd-semantic-segmentation.  Each function cites the
reference file:line whose behaviour it restates.  The reference itself only
delegates to torch ops (torch 1.7.1 ATen, not vendored in /root/reference), so
the op semantics restated here are the published torch semantics
(conv2d, upsample_bilinear2d(align_corners=True), CrossEntropyLoss(ignore_index),
optim.SGD); they are pinned by tests/golden/*.npz, which were produced by running
the reference's own modules in the build container (oracle/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  All functions compute in float64 unless told otherwise, so they
serve as the exact-math reference for kernels that take bf16 operands and
accumulate in fp32.

Layout convention here: NCHW like the reference.
"""
import numpy as np


# --------------------------------------------------------------------------- A5
def frozen_bn_scale_bias(weight, bias, running_mean, running_var):
    """reference core/components/layers.py:18-20: scale = w * rsqrt(var) (NO eps),
    bias' = b - mean * scale."""
    scale = weight / np.sqrt(running_var)
    return scale, bias - running_mean * scale


def frozen_bn(x, weight, bias, running_mean, running_var):
    """reference core/components/layers.py:18-23."""
    s, b = frozen_bn_scale_bias(weight, bias, running_mean, running_var)
    return x * s.reshape(1, -1, 1, 1) + b.reshape(1, -1, 1, 1)


# --------------------------------------------------------------------------- A2
def _out_size(n, k, stride, pad, dil):
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1


def conv2d(x, w, bias=None, stride=1, pad=0, dil=1):
    """nn.Conv2d forward as used by reference core/components/resnet.py:22-30,137
    (conv3x3: pad = dil, no bias; conv1x1; 7x7 stem) and
    core/models/classifiers/aspp/classifier.py:12-20 (pad = dil = rate, bias).
    x [B,C,H,W], w [O,C,kh,kw] -> [B,O,Ho,Wo]; cross-correlation (no flip)."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    B, C, H, W = x.shape
    O, _, kh, kw = w.shape
    Ho, Wo = _out_size(H, kh, stride, pad, dil), _out_size(W, kw, stride, pad, dil)
    xp = np.zeros((B, C, H + 2 * pad, W + 2 * pad), np.float64)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    y = np.zeros((B, O, Ho, Wo), np.float64)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, :, ky * dil: ky * dil + (Ho - 1) * stride + 1: stride,
                       kx * dil: kx * dil + (Wo - 1) * stride + 1: stride]
            y += np.einsum("bchw,oc->bohw", patch, w[:, :, ky, kx], optimize=True)
    if bias is not None:
        y += np.asarray(bias, np.float64).reshape(1, -1, 1, 1)
    return y


def conv2d_dgrad(dy, w, in_hw, stride=1, pad=0, dil=1):
    """d loss / d x for conv2d above (what autograd's convolution_backward returns)."""
    dy = np.asarray(dy, np.float64)
    w = np.asarray(w, np.float64)
    B, O, Ho, Wo = dy.shape
    _, C, kh, kw = w.shape
    H, W = in_hw
    dxp = np.zeros((B, C, H + 2 * pad, W + 2 * pad), np.float64)
    for ky in range(kh):
        for kx in range(kw):
            dxp[:, :, ky * dil: ky * dil + (Ho - 1) * stride + 1: stride,
                kx * dil: kx * dil + (Wo - 1) * stride + 1: stride] += \
                np.einsum("bohw,oc->bchw", dy, w[:, :, ky, kx], optimize=True)
    return dxp[:, :, pad:pad + H, pad:pad + W]


def conv2d_wgrad(dy, x, ksize, stride=1, pad=0, dil=1):
    """d loss / d w for conv2d above."""
    dy = np.asarray(dy, np.float64)
    x = np.asarray(x, np.float64)
    B, O, Ho, Wo = dy.shape
    _, C, H, W = x.shape
    kh = kw = ksize
    xp = np.zeros((B, C, H + 2 * pad, W + 2 * pad), np.float64)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    dw = np.zeros((O, C, kh, kw), np.float64)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, :, ky * dil: ky * dil + (Ho - 1) * stride + 1: stride,
                       kx * dil: kx * dil + (Wo - 1) * stride + 1: stride]
            dw[:, :, ky, kx] = np.einsum("bohw,bchw->oc", dy, patch, optimize=True)
    return dw


# --------------------------------------------------------------------------- A1
def aspp_head(x, weights, biases, rates=(6, 12, 18, 24)):
    """reference core/models/classifiers/aspp/classifier.py:26-29:
    out = conv_0(x); out += conv_i(x) for i = 1..3 (left-to-right sum)."""
    out = conv2d(x, weights[0], biases[0], 1, rates[0], rates[0])
    for i in range(1, len(rates)):
        out = out + conv2d(x, weights[i], biases[i], 1, rates[i], rates[i])
    return out


def aspp_head_backward(dout, x, weights, rates=(6, 12, 18, 24)):
    dout = np.asarray(dout, np.float64)
    dx = np.zeros(np.shape(x), np.float64)
    dws, dbs = [], []
    for i, r in enumerate(rates):
        dx += conv2d_dgrad(dout, weights[i], np.shape(x)[-2:], 1, r, r)
        dws.append(conv2d_wgrad(dout, x, 3, 1, r, r))
        dbs.append(dout.sum((0, 2, 3)))
    return dx, dws, dbs


# --------------------------------------------------------------------------- A3
def _ac_coords(n_in, n_out):
    """align_corners=True source coordinates exactly as ATen computes them in
    float32: scale = (in-1)/(out-1) (0 if out==1); src = scale*dst;
    i0 = floor(src); i1 = min(i0+1, in-1); lambda = src - i0."""
    scale = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
    src = (scale * np.arange(n_out, dtype=np.float32)).astype(np.float32)
    i0 = np.minimum(src.astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    lam = (src - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, lam


def bilinear_ac(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) as called at
    reference classifier.py:31 and core/utils/utility.py:185."""
    x = np.asarray(x, np.float64)
    Ho, Wo = size
    y0, y1, ly = _ac_coords(x.shape[2], Ho)
    x0, x1, lx = _ac_coords(x.shape[3], Wo)
    ly = ly.astype(np.float64).reshape(1, 1, Ho, 1)
    lx = lx.astype(np.float64).reshape(1, 1, 1, Wo)
    top = x[:, :, y0][:, :, :, x0] * (1 - lx) + x[:, :, y0][:, :, :, x1] * lx
    bot = x[:, :, y1][:, :, :, x0] * (1 - lx) + x[:, :, y1][:, :, :, x1] * lx
    return top * (1 - ly) + bot * ly


def bilinear_ac_backward(dy, in_hw):
    dy = np.asarray(dy, np.float64)
    B, C, Ho, Wo = dy.shape
    Hi, Wi = in_hw
    y0, y1, ly = _ac_coords(Hi, Ho)
    x0, x1, lx = _ac_coords(Wi, Wo)
    ly = ly.astype(np.float64)
    lx = lx.astype(np.float64)
    # rows first
    tmp = np.zeros((B, C, Hi, Wo), np.float64)
    np.add.at(tmp, (slice(None), slice(None), y0), dy * (1 - ly).reshape(1, 1, Ho, 1))
    np.add.at(tmp, (slice(None), slice(None), y1), dy * ly.reshape(1, 1, Ho, 1))
    dx = np.zeros((B, C, Hi, Wi), np.float64)
    np.add.at(dx, (slice(None), slice(None), slice(None), x0), tmp * (1 - lx).reshape(1, 1, 1, Wo))
    np.add.at(dx, (slice(None), slice(None), slice(None), x1), tmp * lx.reshape(1, 1, 1, Wo))
    return dx


# --------------------------------------------------------------------------- A4
def log_softmax(x, axis=1):
    x = np.asarray(x, np.float64)
    m = x.max(axis=axis, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(axis=axis, keepdims=True))


def cross_entropy_ignore(logits, labels, ignore_index=255):
    """torch.nn.CrossEntropyLoss(ignore_index=255), reference
    core/trainers/aspp_trainer.py:61,91: mean over non-ignored pixels of
    -log_softmax(x)[y].  Returns (loss, dlogits, n_valid).  n_valid == 0 gives
    nan loss (0/0) and zero gradient-with-nan semantics like torch; we return nan, zeros."""
    logits = np.asarray(logits, np.float64)
    labels = np.asarray(labels).astype(np.int64)
    B, C, H, W = logits.shape
    lsm = log_softmax(logits, 1)
    valid = labels != ignore_index
    n = int(valid.sum())
    safe = np.where(valid, labels, 0)
    picked = np.take_along_axis(lsm, safe[:, None], axis=1)[:, 0]
    if n == 0:
        return float("nan"), np.zeros_like(logits), 0
    loss = -(picked * valid).sum() / n
    onehot = np.zeros_like(logits)
    np.put_along_axis(onehot, safe[:, None], 1.0, axis=1)
    d = (np.exp(lsm) - onehot) * valid[:, None] / n
    return float(loss), d, n


def softmax(x, axis=1):
    return np.exp(log_softmax(x, axis))


# --------------------------------------------------------------------------- A7
def poly_lr(base_lr, it, max_iter, power=0.9):
    """reference core/utils/adapt_lr.py:12-17 ('poly')."""
    return base_lr * ((1 - float(it) / max_iter) ** power)


def sgd_step(p, g, buf, lr, momentum=0.9, weight_decay=5e-4, dtype=np.float32):
    """torch.optim.SGD (dampening 0, no nesterov) as configured at reference
    core/trainers/aspp_trainer.py:25-26: g += wd*p; buf = g (first step) or
    mu*buf + g; p -= lr*buf.  Computed in `dtype` like torch does."""
    p = np.asarray(p, dtype)
    g = np.asarray(g, dtype) + dtype(weight_decay) * p
    buf = g.copy() if buf is None else (dtype(momentum) * np.asarray(buf, dtype) + g)
    return (p - dtype(lr) * buf).astype(dtype), buf.astype(dtype)


# --------------------------------------------------------------------------- A8
def inference_probs(lowres_logits, size):
    """reference core/utils/utility.py:179-191 with flip=False: upsample the
    1/8-resolution logits to the LABEL size, softmax over classes, keep image 0."""
    up = bilinear_ac(lowres_logits, size)
    return softmax(up, 1)[0:1]


# --------------------------------------------------------------------------- A9
def intersection_and_union(pred, target, K, ignore_index=255):
    """reference core/utils/utility.py:148-161 (intersectionAndUnionGPU) and
    :133-145 (numpy twin): pred[target==ignore]=ignore; K-bin histograms of
    pred∩target, pred, target over [0, K-1]; values outside are dropped."""
    pred = np.asarray(pred).reshape(-1).astype(np.int64).copy()
    target = np.asarray(target).reshape(-1).astype(np.int64)
    pred[target == ignore_index] = ignore_index
    inter = pred[pred == target]
    hist = lambda a: np.bincount(a[(a >= 0) & (a < K)], minlength=K).astype(np.float64)
    ai, ao, at = hist(inter), hist(pred), hist(target)
    return ai, ao + at - ai, at, ao


def confusion_matrix(pred, target, K, ignore_label=255):
    """reference core/utils/utility.py:347-359: cmt[gt, pd] += 1 for gt != 255
    (the reference does it with a per-pixel Python loop; same integers)."""
    pred = np.asarray(pred).reshape(-1).astype(np.int64)
    target = np.asarray(target).reshape(-1).astype(np.int64)
    keep = target != ignore_label
    return np.bincount(target[keep] * K + pred[keep], minlength=K * K).reshape(K, K)


class MeterRef:
    """reference core/utils/utility.py:24-72 (AverageMeter) restated."""

    def __init__(self):
        self.i = self.u = self.t = self.r = 0.0
        self.n = 0
        self.iou_sum = 0.0
        self.f1_sum = 0.0

    def update(self, inter, union, target, res):
        self.iou_sum = self.iou_sum + inter / (union + 1e-10)
        self.f1_sum = self.f1_sum + 2 * inter / (target + res + 1e-10)
        self.i, self.u, self.t, self.r = self.i + inter, self.u + union, self.t + target, self.r + res
        self.n += 1

    def summary(self):
        macro_iou, macro_f1 = self.iou_sum / self.n, self.f1_sum / self.n
        micro_iou = self.i / (self.u + 1e-10)
        micro_f1 = 2 * self.i / (self.t + self.r + 1e-10)
        return dict(macro_miou=float(np.mean(macro_iou)), macro_mf1=float(np.mean(macro_f1)),
                    micro_miou=float(np.mean(micro_iou)), micro_mf1=float(np.mean(micro_f1)),
                    macro_iou=macro_iou, micro_iou=micro_iou)
