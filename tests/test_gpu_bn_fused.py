"""Trainable BatchNorm2d, hand-written schedule (round 3): mi_bn_finalize against torch.nn.BatchNorm2d's statistics and running buffers, and the
stem node (conv -> batch statistics -> normalise + ReLU + max-pool in one pass) against torch fp32 composed from the same ops
(resnet.py:137-140, 177-180 with feature_extractor.py:37).  The whole-net checks are tests/test_gpu_bn.py's."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relmax(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("M,C,count_ranks", [(4 * 33 * 29, 64, 1), (2 * 17 * 13, 256, 2), (75272, 1024, 1)])
def test_finalize_matches_torch_batchnorm_statistics_and_running_buffers(M, C, count_ranks):
    from rnd_semantic_segmentation_amd import kernels as K
    g = torch.Generator().manual_seed(M + C)
    y = (torch.randn(M, C, generator=g) * (0.5 + torch.rand(C, generator=g)) + 3.0 * torch.randn(C, generator=g)).to(torch.bfloat16)
    bn = torch.nn.BatchNorm2d(C).to(DEV)
    ref = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        for b in (bn, ref):
            b.weight.copy_(torch.linspace(0.5, 1.5, C))
            b.bias.copy_(torch.linspace(-0.3, 0.3, C))
            b.running_mean.copy_(torch.linspace(-1.0, 1.0, C))
            b.running_var.copy_(torch.linspace(0.5, 2.0, C))
    yd = y.to(DEV).view(1, 1, M, C)
    s1, s2 = K.bn_colsum2(yd, bn.running_mean)
    # `count_ranks` ranks with the same data: sums and count scale together, mean / biased variance stay, the unbiased factor uses the global count
    fin = K.bn_finalize(s1 * count_ranks, s2 * count_ranks, bn.running_mean, M * count_ranks, bn)
    yy = y.double().t().reshape(1, C, M, 1).repeat(count_ranks, 1, 1, 1)
    ref.train()
    out = ref(yy)
    mean, var = yy.mean((0, 2, 3)), yy.var((0, 2, 3), unbiased=False)
    invstd = torch.rsqrt(var + ref.eps)
    # the pilot (running mean, here up to 10 standard deviations off the batch mean - far worse than training ever sees) costs
    # var = E[d^2] - E[d]^2 a few of fp32's 24 bits: measured 4e-5 on invstd in this adversarial case
    assert relmax(fin[0].cpu(), mean) < 2e-6 and relmax(fin[1].cpu(), invstd) < 2e-4
    assert relmax(fin[2].cpu(), ref.weight.detach() * invstd) < 2e-4
    assert float((fin[3].cpu().double() - (ref.bias.detach() - mean * ref.weight.detach() * invstd)).abs().max()) < 2e-4 * float((mean * invstd).abs().max() + 1)
    assert relmax(bn.running_mean.cpu(), ref.running_mean) < 2e-6 and relmax(bn.running_var.cpu(), ref.running_var) < 2e-4
    assert int(bn.num_batches_tracked) == 1
    # the normalise pass with these vectors reproduces torch's output
    got = K.bn_apply(yd, fin[0], fin[2], bn.bias.detach())
    assert relmax(got.view(M, C).cpu(), out[0, :, :, 0].t()) < 2.0 ** -8


def test_stem_node_with_batch_statistics_vs_torch_fp32():
    from rnd_semantic_segmentation_amd.host import engine
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 65, 81
    x = torch.randn(B, 3, H, W, generator=g).to(torch.bfloat16)
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.05)
    bn = torch.nn.BatchNorm2d(64).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, 64))
        bn.bias.copy_(torch.linspace(-0.3, 0.3, 64))
    wd = torch.nn.Parameter(w.to(DEV))
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    pool = engine.BnStemFn.apply(xd, wd, bn.weight, bn.bias, bn)                 # [B,Hp,Wp,64] NHWC bf16
    gout = torch.randn(pool.shape, generator=g).to(torch.bfloat16)
    pool.backward(gout.to(DEV))
    # torch fp32 on the bf16-rounded operands
    ref_bn = torch.nn.BatchNorm2d(64)
    with torch.no_grad():
        ref_bn.weight.copy_(bn.weight.cpu())
        ref_bn.bias.copy_(bn.bias.cpu())
    wr = w.to(torch.bfloat16).float().requires_grad_(True)
    yr = F.conv2d(x.float(), wr, None, 2, 3)
    pr = F.max_pool2d(F.relu(ref_bn(yr)), 3, 2, 1)
    pr.backward(gout.float().permute(0, 3, 1, 2))
    assert relmax(pool.float().cpu().permute(0, 3, 1, 2), pr.detach()) < 3e-2           # y is stored in bf16 before it is normalised
    assert relmax(bn.running_mean.cpu(), ref_bn.running_mean) < 1e-2 and relmax(bn.running_var.cpu(), ref_bn.running_var) < 1e-2
    cos = lambda a, b: float(torch.dot(a.flatten().double(), b.flatten().double()) / (a.double().norm() * b.double().norm()))
    for got, want, name in ((wd.grad.cpu(), wr.grad, "conv1.weight"), (bn.weight.grad.cpu(), ref_bn.weight.grad, "bn1.weight"), (bn.bias.grad.cpu(), ref_bn.bias.grad, "bn1.bias")):
        c = cos(got, want)
        ratio = float(got.double().norm() / want.double().norm())
        print("[bn stem] %s: 1-cos %.3e, norm ratio %.4f" % (name, 1 - c, ratio))
        assert 1 - c < 2e-2 and abs(ratio - 1) < 5e-2, name
