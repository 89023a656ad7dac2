"""First piece of the PraNet path (reference core/trainers/pranet_trainer.py): the structure loss as an autograd
function on the HIP kernel.  The network itself (Res2Net-50 v1b 26w x 4s, RFB, partial decoder, reverse attention) is not built yet:
`core.trainers.pranet_trainer` keeps raising ImportError in the drop-in layer; this module is what its trainer will call."""
import torch

from .. import kernels as K


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
