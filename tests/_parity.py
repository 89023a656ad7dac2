"""Shared helpers of the network-level GPU parity tests (PraNet, GALD): error measures, oracle / engine runners, and the TEACHER-FORCED
comparison.

Why teacher forcing.  Both networks normalise with BatchNorm2d on batch statistics and are 50-80 convolutions deep; run freely in the bf16 regime,
rounding accumulates from block to block - the reference's OWN modules under torch.autocast(bfloat16) end 0.15-0.25 away from their fp32 outputs and
0.3-0.4 (median 1 - cos) away from their fp32 parameter gradients at 8 x 3 x 160 x 160 / 4 x 3 x 352 x 352 with formula weights (DESIGN section 2:
bigger batches and maps, damped residual branches and shifted BatchNorm biases were tried; none brings that below 0.1).  A free-running whole-net
comparison can therefore only be held to that yardstick.  The teacher-forced pass removes the accumulation: every tapped block receives the
ORACLE's fp32 activation (rounded to bf16) at its input and, in backward, the oracle's gradient at its output, so what is compared is one block
deep everywhere - output, gradient handed upstream, and every parameter gradient of the net, at bars around 1e-2 - and an error is attributed to
the block that makes it.
"""
import numpy as np
import torch


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rel2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))


def live(grads, frac=1e-3):
    """Names of the gradient tensors that carry more than `frac` of the largest one (tensors whose exact gradient is zero - a conv bias in front of
    BatchNorm, the key bias of an attention - hold rounding noise only)."""
    gmax = max(float(np.linalg.norm(v)) for v in grads.values())
    return [k for k, v in grads.items() if float(np.linalg.norm(v)) > frac * gmax]


def oracle_run(mods, forward, loss_of, autocast=False):
    """Run the oracle modules: forward() -> tuple of outputs, loss_of(outputs) -> scalar.  Returns (outputs, parameter gradients by
    '<i>.<name>', taps, tap gradients); the modules' state (running statistics) is restored afterwards."""
    sds = [{k: v.clone() for k, v in m.state_dict().items()} for m in mods]
    for m in mods:
        m.zero_grad()
        m.__dict__["_taps"] = {}
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            outs = forward()
    else:
        outs = forward()
    outs = [o.float() for o in outs]
    loss_of(outs).backward()
    taps, tgrads = {}, {}
    for m in mods:
        for k, t in m.__dict__["_taps"].items():
            taps[k] = t.detach().float()
            if t.grad is not None:
                tgrads[k] = t.grad.detach().float()
        m.__dict__["_taps"] = None
    pg = {"%d.%s" % (i, k): p.grad.detach().numpy().copy() for i, m in enumerate(mods) for k, p in m.named_parameters() if p.grad is not None}
    for m, sd in zip(mods, sds):
        m.load_state_dict(sd)
    return [o.detach().numpy() for o in outs], pg, taps, tgrads


def engine_forced(mods, forward, loss_of, taps, tgrads):
    """The engine modules with every tapped activation / gradient replaced by the oracle's.  Returns (the engine's OWN activation at every tap - what
    the producing block made of forced inputs, its own accumulated gradient at every tap, parameter gradients)."""
    own, gown = {}, {}
    for m in mods:
        m.zero_grad()
        m._taps, m._gtaps = {}, {}
        m._force = {k: v.cuda() for k, v in taps.items()}
        m._force_grad = {k: v.cuda() for k, v in tgrads.items()}
    try:
        outs = [o.float() for o in forward()]
        loss_of(outs).backward()
        torch.cuda.synchronize()
        for m in mods:
            for k, v in m._taps.items():
                t = v.t
                own[k] = (t.permute(0, 3, 1, 2) if t.dim() == 4 else t).float().cpu().numpy()
            for k, g in m._gtaps.items():
                gown[k] = (g.permute(0, 3, 1, 2) if g.dim() == 4 else g).float().cpu().numpy()
    finally:
        for m in mods:
            m._taps = m._gtaps = m._force = m._force_grad = None
    pg = {"%d.%s" % (i, k): p.grad.detach().cpu().numpy().copy() for i, m in enumerate(mods) for k, p in m.named_parameters() if p.grad is not None}
    return own, gown, pg


def forced_report(tag, own, gown, pg, taps, tgrads, want_pg, skip=()):
    """Per-block errors of a teacher-forced run: activation (relative L2 against the oracle's), gradient handed upstream (relative L2), and for every
    live parameter tensor |grad| ratio - 1 and 1 - cos.  Prints the worst of each and returns them."""
    act = {k: rel2(own[k], taps[k].numpy()) for k in own if k in taps}
    grd = {k: rel2(gown[k], tgrads[k].numpy()) for k in gown if k in tgrads}
    names = [k for k in live(want_pg) if k in pg and not any(s in k for s in skip)]
    nrm = {k: abs(float(np.linalg.norm(pg[k]) / np.linalg.norm(want_pg[k])) - 1) for k in names}
    dirn = {k: 1 - cos(pg[k], want_pg[k]) for k in names}
    worst = lambda d: max(d.items(), key=lambda kv: kv[1]) if d else ("-", 0.0)
    med = lambda d: float(np.median(list(d.values()))) if d else 0.0
    print("\n[%s forced] %d taps: activation worst %.2e (%s) median %.2e;  upstream gradient worst %.2e (%s) median %.2e\n"
          "[%s forced] %d parameter tensors: |grad| worst %.2e (%s) median %.2e;  1-cos worst %.2e (%s) median %.2e" % (
              tag, len(act), worst(act)[1], worst(act)[0], med(act), worst(grd)[1], worst(grd)[0], med(grd),
              tag, len(names), worst(nrm)[1], worst(nrm)[0], med(nrm), worst(dirn)[1], worst(dirn)[0], med(dirn)))
    return dict(act=act, grd=grd, nrm=nrm, dirn=dirn, act_worst=worst(act)[1], grd_worst=worst(grd)[1], nrm_worst=worst(nrm)[1], dirn_worst=worst(dirn)[1],
                act_median=med(act), grd_median=med(grd), nrm_median=med(nrm), dirn_median=med(dirn), missing=[k for k in live(want_pg) if k not in pg])
