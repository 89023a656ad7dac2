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


@pytest.mark.parametrize("M,C,count_ranks,far", [(4 * 33 * 29, 64, 1, False), (2 * 17 * 13, 256, 2, False), (75272, 1024, 1, False), (75272, 256, 1, True)])
def test_finalize_matches_torch_batchnorm_statistics_and_running_buffers(M, C, count_ranks, far):
    from rnd_semantic_segmentation_amd import kernels as K
    g = torch.Generator().manual_seed(M + C)
    y = (torch.randn(M, C, generator=g) * (0.5 + torch.rand(C, generator=g)) + 3.0 * torch.randn(C, generator=g)).to(torch.bfloat16)
    bn = torch.nn.BatchNorm2d(C).to(DEV)
    ref = torch.nn.BatchNorm2d(C).double()
    # the pilot of the one-pass variance is the running mean: within a standard deviation of the batch mean in training (far: up to ten off)
    rm0 = torch.linspace(-1.0, 1.0, C) if far else y.float().mean(0) + 0.5 * torch.randn(C, generator=g)
    with torch.no_grad():
        for b in (bn, ref):
            b.weight.copy_(torch.linspace(0.5, 1.5, C))
            b.bias.copy_(torch.linspace(-0.3, 0.3, C))
            b.running_mean.copy_(rm0)
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
    # a pilot ten standard deviations off the batch mean (far worse than training sees) costs var = E[d^2] - E[d]^2 several of fp32's 24 bits:
    # measured 5e-4 on invstd at M = 75 272; with the pilot within a standard deviation 2e-6
    tol = 2e-3 if far else 2e-5
    assert relmax(fin[0].cpu(), mean) < 4e-6 and relmax(fin[1].cpu(), invstd) < tol
    assert relmax(fin[2].cpu(), ref.weight.detach() * invstd) < tol
    assert float((fin[3].cpu().double() - (ref.bias.detach() - mean * ref.weight.detach() * invstd)).abs().max()) < tol * float((mean * invstd).abs().max() + 1)
    assert relmax(bn.running_mean.cpu(), ref.running_mean) < 4e-6 and relmax(bn.running_var.cpu(), ref.running_var) < tol
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


@pytest.mark.parametrize("B,H,W,Ca,N,k,stride,dil", [
    (2, 33, 29, 64, 64, 1, 1, 1),          # 128-wide loop, partial tiles in both directions
    (2, 33, 29, 64, 256, 3, 1, 2),         # 3x3 dilated, short contraction
    (2, 41, 37, 256, 128, 1, 2, 1),        # strided (general gather)
    (8, 97, 97, 256, 256, 3, 1, 2),        # BASELINE layer3 conv2: the wide ping-pong loop
    (8, 97, 97, 1024, 256, 1, 1, 1),       # layer3 conv1 (ping-pong, 1x1)
    (8, 97, 97, 256, 1024, 1, 1, 1),       # layer3 conv3 (128-wide loop, HBM-bound)
    (1, 25, 23, 512, 2048, 1, 1, 1),       # N at the reduction's limit
])
def test_conv_epilogue_statistics_equal_a_pass_over_the_output(B, H, W, Ca, N, k, stride, dil):
    """mi_conv_gemm_stats: the output is bit-equal to mi_conv_gemm's, and the sums equal mi_bn_colsum2 of that output up to fp32 summation order
    (both are fixed-order and bitwise reproducible on their own)."""
    from rnd_semantic_segmentation_amd import kernels as K
    g = torch.Generator().manual_seed(B * H + Ca + N + k)
    x = torch.randn(B, H, W, Ca, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, Ca, k, k, generator=g) / (Ca * k * k) ** 0.5).to(DEV)
    wp = K.pack_weight_fwd(w)
    pad = dil * (k // 2)
    Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    pilot = (0.3 * torch.randn(N, generator=g)).to(DEV)
    want = K.conv_gemm(x, wp, (Ho, Wo), k, stride, pad, dil, K.GATHER_FWD)
    got, (s1, s2), _ = K.conv_gemm_stats(x, wp, (Ho, Wo), k, stride, pad, dil, pilot)
    assert torch.equal(got, want)
    r1, r2 = K.bn_colsum2(want, pilot)
    M = B * Ho * Wo
    tol = 4e-6 * (M ** 0.5)
    d = want.float().reshape(M, N) - pilot
    assert float((s1 - r1).abs().max()) < tol * float(d.abs().sum(0).max()) / M ** 0.5 + 1e-3
    assert relmax(s2, r2) < 2e-5
    ref1, ref2 = d.double().sum(0), (d.double() ** 2).sum(0)
    assert float((s1.double() - ref1).abs().max()) <= 2 * float((r1.double() - ref1).abs().max()) + 1e-4 * float(ref1.abs().max())
    assert relmax(s2, ref2) < 1e-5
    again = K.conv_gemm_stats(x, wp, (Ho, Wo), k, stride, pad, dil, pilot)
    assert torch.equal(again[1][0], s1) and torch.equal(again[1][1], s2)
    # finalize fused into the last reduction launch == mi_bn_finalize on the returned sums, running statistics included
    bns = [torch.nn.BatchNorm2d(N).to(DEV) for _ in range(2)]
    for b in bns:
        with torch.no_grad():
            b.weight.copy_(torch.linspace(0.5, 1.5, N))
            b.bias.copy_(torch.linspace(-0.3, 0.3, N))
            b.running_mean.copy_(pilot)
    fused = K.conv_gemm_stats(x, wp, (Ho, Wo), k, stride, pad, dil, bns[0].running_mean, bn=bns[0])
    fin = K.bn_finalize(s1, s2, bns[1].running_mean, M, bns[1])
    assert torch.equal(fused[2], fin) and torch.equal(bns[0].running_mean, bns[1].running_mean) and torch.equal(bns[0].running_var, bns[1].running_var)
    assert int(bns[0].num_batches_tracked) == 1
