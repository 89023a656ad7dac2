"""Architecture tables for the dilated ResNet backbone (output stride 8) and helpers to hang
parameters on an nn.Module tree so that state_dict keys equal the reference's.

Reference: core/components/resnet.py:118-191 (ResNet.__init__/_make_layer with
replace_stride_with_dilation=[False, True, True] from core/models/feature_extractor.py:42),
core/components/resnet.py:73-113 (Bottleneck), torchvision IntermediateLayerGetter keeping
children up to layer4 under the `backbone.` prefix (feature_extractor.py:45-52).
"""
from collections import namedtuple

import torch.nn as nn

LAYERS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3)}

Block = namedtuple("Block", "name cin width cout stride dil down")
Conv = namedtuple("Conv", "key bn cout cin k stride pad dil")


def bottleneck_plan(layers, dilate=(False, True, True)):
    """One row per bottleneck.  A stage whose stride is replaced by dilation keeps the previous dilation in its
    first block and uses the multiplied dilation in the rest (resnet.py:171-190)."""
    rows = []
    cin, dil = 64, 1
    for stage, (width, nblk) in enumerate(zip((64, 128, 256, 512), layers)):
        stride = 1 if stage == 0 else 2
        first_dil = dil
        if stage > 0 and dilate[stage - 1]:
            dil, stride = dil * stride, 1
        for b in range(nblk):
            head = b == 0
            rows.append(Block("layer%d.%d" % (stage + 1, b), cin, width, width * 4, stride if head else 1,
                              first_dil if head else dil, head and (stride != 1 or cin != width * 4)))
            cin = width * 4
    return rows


def block_convs(blk):
    """The (up to) four convs of a bottleneck with their FrozenBN path: conv1 1x1, conv2 3x3 (stride, dilation,
    pad = dilation: resnet.py:22-25), conv3 1x1, downsample 1x1 (stride)."""
    n = blk.name
    out = [Conv(n + ".conv1", n + ".bn1", blk.width, blk.cin, 1, 1, 0, 1),
           Conv(n + ".conv2", n + ".bn2", blk.width, blk.width, 3, blk.stride, blk.dil, blk.dil),
           Conv(n + ".conv3", n + ".bn3", blk.cout, blk.width, 1, 1, 0, 1)]
    if blk.down:
        out.append(Conv(n + ".downsample.0", n + ".downsample.1", blk.cout, blk.cin, 1, blk.stride, 0, 1))
    return out


STEM = Conv("conv1", "bn1", 64, 3, 7, 2, 3, 1)


class Holder(nn.Module):
    """Bare container node; lets `backbone.layer3.7.conv2.weight` resolve like in the reference."""


def node_at(root, dotted):
    cur = root
    for part in dotted.split("."):
        if not hasattr(cur, part):
            cur.add_module(part, Holder())
        cur = getattr(cur, part)
    return cur


def out_hw(h, w, conv):
    f = lambda n: (n + 2 * conv.pad - conv.dil * (conv.k - 1) - 1) // conv.stride + 1
    return f(h), f(w)
