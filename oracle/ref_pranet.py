"""ORACLE (test infrastructure - never imported by the product path): CPU restatement of the PraNet pieces of SURVEY 8f row N3,
pinned by fixtures generated from the reference's own code (oracle/make_golden.py: g11_*).

  structure_loss     reference core/trainers/pranet_trainer.py:22-31
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
