"""GPU parity of the first kernels of the GALD / GCPA path (csrc/gald.hip) through the C-ABI: the depthwise 3x3 conv
(forward with BatchNorm tile statistics, data / weight / bias gradient) against torch in float64, and the reference's CrissCrossAttention
(contextagg/ccnet.py:37-127) and LocalAttenModule (contextagg/GALDNet.py:124-157) COMPOSED from the kernels - forward and backward -
against the reference's own fp32 run (g13_gald_modules), with the oracle's tensors for the gradient directions.  bf16 operands and
activations, fp32 accumulation: outputs within 3x the measured deviation (comment next to each bar)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import _cases
from rnd_semantic_segmentation_amd.host import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gk():
    import __graft_entry__ as entry
    entry.build()
    from rnd_semantic_segmentation_amd import gk as g
    return g


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def rel2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _bf(t):
    return t.to(torch.bfloat16)


def _u(name, shape, s=1.0):
    return (synth.uniform("gald." + name, shape) * s).astype(np.float32)


@pytest.mark.parametrize("C,H,W,stride,pad", [(32, 19, 17, 2, 0), (256, 11, 11, 2, 0), (40, 9, 12, 1, 1)])
def test_depthwise_conv_forward_statistics_and_gradients(gk, C, H, W, stride, pad):
    g = torch.Generator().manual_seed(C + H)
    x = _bf(torch.randn((2, C, H, W), generator=g))
    w = torch.randn((C, 1, 3, 3), generator=g) * 0.4
    b = torch.randn(C, generator=g) * 0.2
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv2d(xd, wd, bd, stride, pad, 1, C)
    dy = _bf(torch.randn(tuple(ref.shape), generator=g))
    ref.backward(dy.double())
    big = torch.full((2, H, W, C + 8), 7.0, dtype=torch.bfloat16, device="cuda")
    big[..., 4:4 + C] = _nhwc(x).cuda()
    xv = big[..., 4:4 + C]                                                   # a channel-slice view
    out, st = gk.gdwconv(xv, w.cuda(), b.cuda(), stride, pad, stats=True)
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).double().cpu()
    assert float((got - ref.detach()).abs().max()) < 2.0 ** -8 * float(ref.abs().max()) + 1e-6
    tiles = st.numel() // (2 * C)
    s = st.view(tiles, 2, C).double().sum(0).cpu()
    o64 = out.double().cpu().reshape(-1, C)
    assert torch.allclose(s[0], o64.sum(0), rtol=1e-5, atol=1e-4) and torch.allclose(s[1], (o64 * o64).sum(0), rtol=1e-5, atol=1e-5)
    dw, db = torch.empty_like(w, device="cuda"), torch.empty(C, device="cuda")
    dx = gk.gdwconv_backward(_nhwc(dy).cuda(), xv, w.cuda(), dw, db, stride, pad)
    torch.cuda.synchronize()
    assert rel(dx.permute(0, 3, 1, 2).float().cpu().numpy(), xd.grad.numpy()) < 2.0 ** -7
    assert rel(dw.cpu().numpy(), wd.grad.numpy()) < 2e-5 and rel(db.cpu().numpy(), bd.grad.numpy()) < 2e-5
    dw2, db2 = torch.empty_like(dw), torch.empty_like(db)
    gk.gdwconv_backward(_nhwc(dy).cuda(), xv, w.cuda(), dw2, db2, stride, pad, need_dx=False)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)                     # fixed-order reduction


def _conv1x1(gk, x, w, b):
    wp, wpt = gk.gconv_pack(w)
    y, _ = gk.gconv(x, wp, w.shape[0], (1, 1, 1, 1, 0, 0, 1, 1), bias=b)
    return y, wpt


def test_criss_cross_attention_composed_from_kernels_vs_reference_golden(gk):
    """CrissCrossAttention(64) on 2 x 64 x 5 x 7 (g13 `cca`): q / k / v projections by the general conv (with bias), the attention core by
    mi_gcca_fwd / mi_gcca_bwd, gamma * agg + x by mi_gbn_apply; backward by hand in the order autograd takes."""
    from oracle import ref_gald as rg
    g = _cases.load("g13_gald_modules")
    x0 = _u("cca.x", (2, 64, 5, 7), 3)
    ref = rg.CrissCross(64)
    synth.load_formula_weights(ref, prefix="cca.")
    P = {k: v.detach().cuda() for k, v in ref.named_parameters()}
    x = _nhwc(_bf(torch.from_numpy(x0))).cuda()
    B, H, W, C = x.shape
    q, wq_t = _conv1x1(gk, x, P["query_conv.weight"], P["query_conv.bias"])
    k, wk_t = _conv1x1(gk, x, P["key_conv.weight"], P["key_conv.bias"])
    v, wv_t = _conv1x1(gk, x, P["value_conv.weight"], P["value_conv.bias"])
    agg, att = gk.gcca_fwd(q, k, v)
    gamma = P["gamma"].expand(C).contiguous()
    out = gk.gbn_apply(agg, gamma, torch.zeros(C, device="cuda"), False, add=x)
    torch.cuda.synchronize()
    assert abs(float(att.sum()) - B * H * W) < 1e-3 and float(att[:, torch.arange(H), :, torch.arange(H)].abs().max()) == 0.0      # rows sum to 1; own column position masked
    e_out = rel(out.permute(0, 3, 1, 2).float().cpu().numpy(), g["cca_out"])
    # the loss of the fixture: mean(y^2) + mean(y)
    n = out.numel()
    dout = _bf((2.0 * out.float() + 1.0) / n)
    dagg = gk.gbn_apply(dout, gamma, torch.zeros(C, device="cuda"), False)
    dgam_c = torch.empty(C, device="cuda")
    gk.gbn_bwd_sums(dout, agg, None, torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.empty(C, device="cuda"), dgam_c)
    dq, dk, dv = gk.gcca_bwd(q, k, v, att, dagg)
    grads, dx = {"gamma": dgam_c.sum().reshape(1)}, dout.float()
    geom = (1, 1, 1, 1, 0, 0, 1, 1)
    for name, d, wt in (("query_conv", dq, wq_t), ("key_conv", dk, wk_t), ("value_conv", dv, wv_t)):
        dw = torch.empty_like(P[name + ".weight"])
        gk.gconv_wgrad(d, x, dw, geom)
        db = torch.empty_like(P[name + ".bias"])
        gk.gbn_bwd_sums(d, None, None, None, None, db, None)
        grads[name + ".weight"], grads[name + ".bias"] = dw, db
        dxi, _ = gk.gconv(d, wt, C, geom, mode=gk.GATHER_DGRAD, out_hw=(H, W))
        dx = dx + dxi.float()
    torch.cuda.synchronize()
    e_dx = rel2(dx.permute(0, 3, 1, 2).cpu().numpy(), g["cca_dx0"])
    # oracle tensors for the directions
    ref.train()
    rx = torch.from_numpy(x0).requires_grad_(True)
    ry = ref(rx)
    (ry.square().mean() + ry.mean()).backward()
    # (key_conv.bias shifts every affinity of a query by the same q . b: the softmax does not see it and its exact gradient is zero - the
    # reference holds 1e-10 of rounding there; tensors below 1e-3 of the largest gradient are left out of the relative comparisons)
    rgd = {k: p.grad for k, p in ref.named_parameters()}
    gmax = max(float(v.norm()) for v in rgd.values())
    live = [k for k, v in rgd.items() if float(v.norm()) > 1e-3 * gmax]
    assert "key_conv.bias" not in live and float(grads["key_conv.bias"].norm()) < 1e-2 * gmax
    worst = max(1 - _cos(grads[k].cpu().numpy(), rgd[k].numpy()) for k in live)
    names = [str(s) for s in g["cca_pnames"]]
    assert np.allclose([float(rgd[k].double().norm()) for k in names], g["cca_pgrad"], rtol=2e-3, atol=1e-7)       # the oracle's gradients are the reference's
    e_norm = max(abs(float(grads[k].double().norm()) / float(rgd[k].double().norm()) - 1) for k in live)
    print("\n[cca] out %.2e  dx %.2e  |grad| %.2e  1-cos %.2e" % (e_out, e_dx, e_norm, worst))
    assert e_out < 1.5e-2 and e_dx < 3e-2 and e_norm < 6e-2 and worst < 6e-3      # bars 3x measured (see the printed line)


def test_local_attention_module_composed_from_kernels_vs_reference_golden(gk):
    """LocalAttenModule(32) on 2 x 32 x 19 x 17 (g13 `lam`): two depthwise stride-2 convs (bias) each with BatchNorm on batch statistics (tile
    statistics from the conv, mi_gbn_finalize / mi_gbn_apply) and ReLU, bilinear align_corners=True back to 19 x 17, sigmoid gate."""
    from oracle import ref_gald as rg
    g = _cases.load("g13_gald_modules")
    x0 = _u("lam.x", (2, 32, 19, 17), 3)
    ref = rg.LocalAtten(32)
    synth.load_formula_weights(ref, prefix="lam.")
    P = {k: v.detach().cuda() for k, v in ref.state_dict().items()}
    x = _nhwc(_bf(torch.from_numpy(x0))).cuda()
    B, H, W, C = x.shape

    def unit(inp, i):
        y, st = gk.gdwconv(inp, P["dconv%d.0.weight" % i], P["dconv%d.0.bias" % i], 2, 0, stats=True)
        M = y.shape[0] * y.shape[1] * y.shape[2]
        fin = gk.gbn_finalize(st, C, M, P["dconv%d.1.weight" % i], P["dconv%d.1.bias" % i], P["dconv%d.1.running_mean" % i].clone(), P["dconv%d.1.running_var" % i].clone(), 0.1, 1e-5)
        return y, fin, gk.gbn_apply(y, fin[2], fin[3], True), M
    y1, f1, a1, M1 = unit(x, 1)
    y2, f2, a2, M2 = unit(a1, 2)
    up = gk.gresize(a2, (H, W), True)
    out = gk.ggate(x, up)
    torch.cuda.synchronize()
    e_out = rel(out.permute(0, 3, 1, 2).float().cpu().numpy(), g["lam_out"])
    dout = _bf((2.0 * out.float() + 1.0) / out.numel())
    dx_gate, dup = gk.ggate_bwd(x, up, dout)
    da2 = gk.gresize_bwd(dup, (a2.shape[1], a2.shape[2]), True)
    grads = {}

    def unit_back(gout, y, fin, act, inp, i, M, need_dx):
        db, dg = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        gk.gbn_bwd_sums(gout, y, act, fin[0], fin[1], db, dg)
        dy = gk.gbn_bwd_apply(gout, y, act, fin[0], fin[1], P["dconv%d.1.weight" % i], db, dg, M)
        dw, dbias = torch.empty_like(P["dconv%d.0.weight" % i]), torch.empty(C, device="cuda")
        dxi = gk.gdwconv_backward(dy, inp, P["dconv%d.0.weight" % i], dw, dbias, 2, 0, need_dx=need_dx)
        grads.update({"dconv%d.1.bias" % i: db, "dconv%d.1.weight" % i: dg, "dconv%d.0.weight" % i: dw, "dconv%d.0.bias" % i: dbias})
        return dxi
    da1 = unit_back(da2, y2, f2, a2, a1, 2, M2, True)
    dx1 = unit_back(da1, y1, f1, a1, x, 1, M1, True)
    dx = dx_gate.float() + dx1.float()
    torch.cuda.synchronize()
    e_dx = rel2(dx.permute(0, 3, 1, 2).cpu().numpy(), g["lam_dx0"])
    ref.train()
    rx = torch.from_numpy(x0).requires_grad_(True)
    ry = ref(rx)
    (ry.square().mean() + ry.mean()).backward()
    rg_ = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
    gmax = max(float(v.norm()) for v in rg_.values())
    live = [k for k, v in rg_.items() if float(v.norm()) > 1e-3 * gmax]          # (a conv bias in front of BatchNorm has a zero gradient)
    worst = max(1 - _cos(grads[k].cpu().numpy(), rg_[k].numpy()) for k in live)
    e_norm = max(abs(float(grads[k].double().norm()) / float(rg_[k].double().norm()) - 1) for k in live)
    print("\n[lam] out %.2e  dx %.2e  |grad| %.2e  1-cos %.2e" % (e_out, e_dx, e_norm, worst))
    assert e_out < 1.5e-2 and e_dx < 6e-2 and e_norm < 0.1 and worst < 2e-2


# ------------------------------------------------------------------------------------------------ the whole network on the HIP engine
def _deviation(got, want):
    o, pg = got
    o0, pg0 = want
    gmax = max(np.linalg.norm(v) for v in pg0.values())
    big = [k for k, v in pg0.items() if np.linalg.norm(v) > 1e-3 * gmax]
    return dict(out=max(rel2(a, b) for a, b in zip(o, o0)), norm=max(abs(np.linalg.norm(pg[k]) / np.linalg.norm(pg0[k]) - 1) for k in big),
                cos=max(1 - _cos(pg[k], pg0[k]) for k in big), cos_median=float(np.median([1 - _cos(pg[k], pg0[k]) for k in big])))


def test_gald_whole_net_224_vs_reference_golden(golden_dir):
    """GCPAEncoder (HarDNet-68) + GCPADecoder on the HIP engine at 2 x 3 x 224 x 224 against the reference's own run (g13_gald_224): state_dict
    keys, feature shapes, the four class-logit outputs, the four cross-entropies and their weighted sum (gald_trainer.py:66-84), every
    parameter gradient (norm and direction against the oracle's tensors) - with the reference's own bf16-autocast run as the yardstick for
    the bf16 regime (tests/test_gpu_pranet.py explains why): the engine may deviate from the fp32 reference at most 2x as far as that run does."""
    import json
    import os
    from oracle import ref_gald as rg
    from rnd_semantic_segmentation_amd.host import gald
    keys = json.load(open(os.path.join(golden_dir, "g8_gald_keys.json")))
    enc, dec = gald.GCPAEncoder(), gald.GCPADecoder()
    renc, rdec = rg.GCPAEncoder(), rg.GCPADecoder()
    assert list(enc.state_dict().keys()) == keys["encoder"] and list(dec.state_dict().keys()) == keys["decoder"]
    for m, r, pre in ((enc, renc, "gald.enc."), (dec, rdec, "gald.dec.")):
        synth.load_formula_weights(m, prefix=pre)
        synth.load_formula_weights(r, prefix=pre)
        m.cuda().train()
        r.train()
    g = _cases.load("g13_gald_224")
    x = torch.from_numpy(synth.synth_image(2, 224, 224, seed=51))
    lab = torch.from_numpy(synth.synth_label(2, 224, 224, 19, seed=51)).long()
    # engine
    crit = gald.CrossEntropyNHWC(255)
    feats = enc(x.cuda())
    assert [list(f.shape) for f in feats] == g["feat_shapes"].tolist()
    outs = dec(x.cuda(), feats)
    ls = [crit(o, lab.cuda()) for o in outs]
    (ls[3] * 1 + ls[2] * 0.8 + ls[1] * 0.6 + ls[0] * 0.4).backward()
    torch.cuda.synchronize()
    ours = ([o.detach().float().cpu().numpy() for o in outs],
            {pre + k: p.grad.detach().cpu().numpy().copy() for pre, m in (("e.", enc), ("d.", dec)) for k, p in m.named_parameters() if p.grad is not None})
    e_loss = float(np.abs(np.array([float(l) for l in ls]) / g["losses"] - 1).max())

    def oracle(autocast):
        for m in (renc, rdec):
            m.zero_grad()
        sd = [{k: v.clone() for k, v in m.state_dict().items()} for m in (renc, rdec)]
        if autocast:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                o = rdec(x, renc(x))
        else:
            o = rdec(x, renc(x))
        o = [t.float() for t in o]
        losses, loss = rg.gald_losses(o, lab)
        loss.backward()
        res = ([t.detach().numpy() for t in o], {pre + k: p.grad.numpy().copy() for pre, m in (("e.", renc), ("d.", rdec)) for k, p in m.named_parameters() if p.grad is not None})
        for m, s in zip((renc, rdec), sd):
            m.load_state_dict(s)
        return res, [float(l) for l in losses]
    want, l32 = oracle(False)
    auto, l16 = oracle(True)
    assert np.allclose(l32, g["losses"], rtol=2e-5)                                      # the oracle's fp32 run IS the reference's
    for i in range(4):
        assert rel(want[0][i][:, :, ::16, ::16], g["out%d_crop" % i]) < 1e-4
    y_loss = float(np.abs(np.array(l16) / g["losses"] - 1).max())
    unused = [k for k in want[1] if k not in ours[1]]
    assert not unused, unused[:5]
    dev, yard = _deviation(ours, want), _deviation(auto, want)
    print("\n[gald 224] losses: engine %.2e, autocast %.2e\n[gald 224] engine   out %.2e |grad| %.2e 1-cos %.2e (median %.2e)\n[gald 224] autocast out %.2e |grad| %.2e 1-cos %.2e (median %.2e)" % (
        e_loss, y_loss, dev["out"], dev["norm"], dev["cos"], dev["cos_median"], yard["out"], yard["norm"], yard["cos"], yard["cos_median"]))
    assert e_loss <= max(2 * y_loss, 2e-2)
    for k, floor in (("out", 2e-2), ("norm", 3e-2), ("cos", 2e-3), ("cos_median", 1e-3)):
        assert dev[k] <= max(2 * yard[k], floor), (k, dev[k], yard[k])
    # dconv3 of the local attention modules and the ImageNet head are never run: no gradient in either implementation
    assert float(dec.local_attention_4.dconv3._modules["0"].weight.grad.abs().max() if dec.local_attention_4.dconv3._modules["0"].weight.grad is not None else 0.0) == 0.0


@pytest.mark.parametrize("h,w,H,W", [(7, 7, 224, 224), (23, 40, 90, 160), (12, 9, 24, 18)])
def test_fused_upsample_cross_entropy_align_corners_false(h, w, H, W):
    """mi_upsample_ce_ex with align_corners = 0 (F.interpolate(size=, mode="bilinear") + CrossEntropyLoss(ignore_index=255), gcpa_cc2.py:78-81 +
    gald_trainer.py:76-79) against torch in float64: loss and the gradient with respect to the low-resolution logits."""
    from rnd_semantic_segmentation_amd import kernels as K
    g = torch.Generator().manual_seed(h * W)
    low = torch.randn((2, 19, h, w), generator=g) * 3
    lab = torch.from_numpy(synth.synth_label(2, H, W, 19, seed=h)).long()
    ld = low.double().requires_grad_(True)
    ref = F.cross_entropy(F.interpolate(ld, size=(H, W), mode="bilinear"), lab, ignore_index=255)
    ref.backward()
    out, dlow = K.upsample_ce(low.permute(0, 2, 3, 1).contiguous().cuda(), lab.cuda(), align_corners=False)
    torch.cuda.synchronize()
    assert abs(float(out[0]) - float(ref)) < 2e-6 * abs(float(ref)) + 1e-7
    assert rel(dlow.permute(0, 3, 1, 2).cpu().numpy(), ld.grad.numpy()) < 2e-5


def test_gald_fused_loss_heads_equal_the_materialised_path():
    """GCPADecoder.losses (upsample + cross-entropy fused, what GALDTrainer runs) against decoder(x, feats) + CrossEntropyNHWC on the materialised
    [B,19,H,W] logits: the four losses and every parameter gradient of encoder and decoder."""
    from rnd_semantic_segmentation_amd.host import gald
    x = torch.from_numpy(synth.synth_image(2, 224, 256, seed=61)).cuda()
    lab = torch.from_numpy(synth.synth_label(2, 224, 256, 19, seed=61)).long().cuda()

    def run(fused):
        torch.manual_seed(3)
        enc, dec = gald.GCPAEncoder().cuda().train(), gald.GCPADecoder().cuda().train()
        with torch.no_grad():
            dec.long_relation.gamma.fill_(0.3)
        feats = enc(x)
        if fused:
            ls = dec.losses(x, feats, lab)
        else:
            crit = gald.CrossEntropyNHWC(255)
            ls = [crit(o, lab) for o in dec(x, feats)]
        (ls[3] * 1 + ls[2] * 0.8 + ls[1] * 0.6 + ls[0] * 0.4).backward()
        torch.cuda.synchronize()
        return [float(l) for l in ls], {k: p.grad.detach().clone() for m in (enc, dec) for k, p in m.named_parameters() if p.grad is not None}
    la, ga = run(True)
    lb, gb = run(False)
    assert np.allclose(la, lb, rtol=2e-6), (la, lb)
    # the data gradient of the heads is rounded to bf16 once in either path, but from different fp32 values (fused: exact sums; materialised: via the
    # fp32 upsampled gradient): directions agree to the bf16 level
    # (tensors whose exact gradient is zero - a conv bias in front of BatchNorm, key_conv.bias - hold rounding noise only: left out)
    gmax = max(float(v.norm()) for v in gb.values())
    worst = max(1 - _cos(ga[k].cpu().numpy(), gb[k].cpu().numpy()) for k in ga if float(gb[k].norm()) > 1e-3 * gmax)
    assert worst < 1e-2, worst
