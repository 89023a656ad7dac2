"""Deterministic, RNG-free synthetic weights / images / labels.

There is no network in the build or on the GPU box, so ImageNet weights
(`MODEL.WEIGHTS` URL in the reference's YAMLs, reference
core/components/resnet.py:211-215) and Cityscapes/GTA5 images are unavailable.
Everything here is a pure function of (tensor name, element index) computed in
integer arithmetic, so the golden-fixture generator (which runs the reference
model in the build container) and the GPU box regenerate bit-identical
tensors without shipping 170 MB of weights.

Tensor contract of the data side follows reference
core/datasets/transform.py:31-46 and core/configs/defaults.py:21-24:
image f32 [3,H,W] already mean/std normalised, label f32 [H,W] holding
train-ids 0..K-1 and 255 (ignore).
"""
import zlib

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x):
    """murmur3 finaliser on uint64 arrays holding 32-bit values."""
    x = x & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M32
    x ^= x >> np.uint64(16)
    return x


def hash_u32(name, n, salt=0):
    """n pseudo-random uint32 values, a pure function of (name, salt, index)."""
    seed = np.uint64((zlib.crc32(name.encode()) ^ (salt * 0x9E3779B1)) & 0xFFFFFFFF)
    idx = np.arange(n, dtype=np.uint64)
    return _mix32(_mix32(idx + seed) ^ (seed * np.uint64(0x27D4EB2F) & _M32))


def uniform(name, shape, salt=0):
    """float64 uniform in [-0.5, 0.5), exact dyadic rationals (k/2^32 - 0.5)."""
    n = int(np.prod(shape))
    u = hash_u32(name, n, salt).astype(np.float64) / 4294967296.0 - 0.5
    return u.reshape(shape)


_SQRT12 = 3.4641016151377544
# The stem FrozenBN scales activations by ACT_SCALE and every BN shift is proportional to it, so the (positively
# homogeneous) ReLU network runs at |feature| ~ 0.8, |logit| ~ 1.  At scale 1 the feature common mode makes the very
# first SGD step at the config's LR (5e-4, classifier x10) overshoot and the loss diverges - in the fp32 reference
# too (measured: 4.9 -> 8.4 -> 15); at 0.125 it decreases monotonically.
ACT_SCALE = 0.125


def formula_tensor(key, shape):
    """Value of state_dict entry `key` (float32 numpy).

    conv weights : uniform, std = sqrt(2 / fan_in)   (keeps post-ReLU second moment ~1)
    ASPP weights : uniform, std = 0.01               (reference classifier.py:23-24 uses N(0,0.01))
    FrozenBN     : weight 0.25 for the last BN of a residual branch, 0.5 for a downsample BN,
                   ACT_SCALE for the stem BN, else 1 (all +-5 %); bias, running_mean small (x ACT_SCALE);
                   running_var in [0.8, 1.2]  -> exercises the no-eps rsqrt of
                   reference core/components/layers.py:18-23 without blowing up.
    """
    shape = tuple(int(s) for s in shape)
    u = uniform(key, shape)
    if key.startswith(("D.", "cls1.", "cls2.")):            # PixelDiscriminator (reference core/models/discriminator.py:31-50)
        if key.endswith(".weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            return (u * _SQRT12 * np.sqrt(1.0 / fan_in)).astype(np.float32)
        return (u * 0.1).astype(np.float32)
    if key.startswith("conv2d_list."):
        if key.endswith(".weight"):
            return (u * _SQRT12 * 0.01).astype(np.float32)
        return (u * 0.2).astype(np.float32)  # bias
    if key.endswith("num_batches_tracked"):
        return np.zeros(shape, np.int64)
    leaf = key.rsplit(".", 1)[-1]
    parent = key.rsplit(".", 1)[0]
    is_bn = parent.rsplit(".", 1)[-1].startswith("bn") or ".downsample.1" in key
    # BatchNorm tensors are 1-D; any 1-D `weight` / running statistic is one whatever its module is called (`norm` in HarDNet's ConvLayer, an index
    # inside a Sequential: resnet.conv1.1, downsample.2 of Res2Net, conva.1, dconvN.1).  The name rules alone had taken Res2Net's 4-D
    # `downsample.1` CONV weight for a BatchNorm gain (every weight ~0.5: all output channels of the shortcut nearly equal - a rank-one trunk that
    # amplified rounding tenfold) and left those BatchNorms with gains of +-0.05 and running variances of either sign (eval(): NaN).
    is_bn = (is_bn or leaf in ("running_mean", "running_var") or leaf == "weight") and len(shape) == 1
    if is_bn:
        stem = ".layer" not in "." + key and parent.endswith("bn1")
        if leaf == "weight":
            base = ACT_SCALE if stem else (0.25 if parent.endswith("bn3") else (0.5 if ".downsample.1" in key else 1.0))
            return (base * (1.0 + 0.1 * u)).astype(np.float32)
        if leaf == "bias":
            return (ACT_SCALE * 0.1 * u).astype(np.float32)
        if leaf == "running_mean":
            return ((1.0 if stem else ACT_SCALE) * 0.2 * u).astype(np.float32)
        if leaf == "running_var":
            return (1.0 + 0.4 * u).astype(np.float32)
    if leaf == "weight" and len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        return (u * _SQRT12 * np.sqrt(2.0 / fan_in)).astype(np.float32)
    if leaf == "weight" and len(shape) == 2:
        return (u * _SQRT12 * np.sqrt(1.0 / shape[1])).astype(np.float32)
    return (0.1 * u).astype(np.float32)


COND_BN_BIAS = 2.0


def load_formula_weights(module, prefix="", bn_bias=0.0):
    """Fill every parameter and buffer of `module` with formula_tensor(prefix+key).

    bn_bias: added to the bias of every BatchNorm (a `.bias` whose sibling `.running_mean` exists).  COND_BN_BIAS is the CONDITIONED regime of the
    whole-network parity fixtures of the BatchNorm-on-batch-statistics nets (PraNet, GALD): with zero-mean pre-activations half of every layer's
    units sit at the ReLU kink, a deep batch-normalised random net is chaotic (rounding doubles per 16-layer HarDBlock: the reference's own
    bf16-autocast run ends 40 % away from its fp32 run, gradient directions 1 - cos = 0.75) and no implementation can be compared with another
    through 70 layers; with the pre-activations centred two standard deviations above the kink (2.3 % of the units inactive) the same nets are
    well-conditioned (autocast: outputs 1-5 %, 1 - cos 1e-3 .. 1e-2) and every conv / link / resize / attention is exercised all the same."""
    import torch

    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        a = formula_tensor(prefix + k, v.shape)
        if bn_bias and k.endswith(".bias") and k[:-5] + ".running_mean" in sd:
            a = (a + np.float32(bn_bias)).astype(np.float32)
        new[k] = torch.from_numpy(a).to(v.dtype)
    module.load_state_dict(new)
    return module


def synth_image(batch, height, width, seed=1234):
    """float32 [B,3,H,W], ~N(0,1)-ish (sum of 4 uniforms), post-normalisation statistics."""
    shape = (batch, 3, height, width)
    acc = np.zeros(shape, np.float64)
    for s in range(4):
        acc += uniform("image", shape, salt=seed * 4 + s)
    return (acc * (_SQRT12 / 2.0)).astype(np.float32)


def synth_label(batch, height, width, num_classes=19, seed=1234, ignore=255, border=None):
    """float32 [B,H,W] train-ids (as the reference loader yields floats,
    core/datasets/cityscapes.py:137-151) in blocky regions, with an ignore
    border band and ~2 % scattered ignore pixels."""
    if border is None:
        border = max(1, min(height, width) // 24)
    cell = max(4, min(height, width) // 12)
    gh, gw = -(-height // cell), -(-width // cell)
    grid = hash_u32("label_grid", batch * gh * gw, salt=seed).reshape(batch, gh, gw) % np.uint64(num_classes)
    lab = np.repeat(np.repeat(grid, cell, axis=1), cell, axis=2)[:, :height, :width].astype(np.float32)
    noise = hash_u32("label_noise", batch * height * width, salt=seed).reshape(batch, height, width)
    lab[(noise % np.uint64(50)) == 0] = ignore
    lab[:, :border, :] = ignore
    lab[:, -border:, :] = ignore
    lab[:, :, :border] = ignore
    lab[:, :, -border:] = ignore
    return lab


def synth_polyp(batch, height, width, seed=1234):
    """(image float32 [B,3,H,W], mask float32 [B,1,H,W] in {0,1}): one or two elliptic blobs per image; the image is the noise of
    synth_image plus a brightness offset inside the blob, so that a network CAN learn the mask (training-parity tests)."""
    img = synth_image(batch, height, width, seed=seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    mask = np.zeros((batch, 1, height, width), np.float32)
    par = uniform("polyp", (batch, 2, 5), salt=seed) + 0.5          # in [0, 1)
    for b in range(batch):
        for k in range(2):
            cy, cx, ry, rx, on = par[b, k]
            if k == 1 and on < 0.5:
                continue
            cy, cx = (0.2 + 0.6 * cy) * height, (0.2 + 0.6 * cx) * width
            ry, rx = (0.08 + 0.17 * ry) * height, (0.08 + 0.17 * rx) * width
            mask[b, 0][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 1.0
    img = img + 1.5 * mask
    return img.astype(np.float32), mask


def bf16_round(a):
    """Round float32 numpy array to the nearest bf16 (ties to even), returned as float32."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)) << np.uint64(16)
    return (r & _M32).astype(np.uint32).view(np.float32).reshape(a.shape)
