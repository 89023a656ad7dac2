"""FADA adversarial domain adaptation around the DeepLabV2 path, on the MI355X engine.

Reference surface mirrored here (same names / arguments / attributes / checkpoint keys):
  PixelDiscriminator            core/models/discriminator.py:31-50
  build_adversarial_discriminator  core/models/build.py:33-53
  soft_label_cross_entropy      core/utils/utility.py:172-177
  FADAAdapter                   core/adapters/fada_adapter.py:6-31
  AsppFada                      core/combos/aspp_fada.py:13-198

Every conv of the discriminator is the same implicit-GEMM kernel as the backbone's (3x3, pad 1; bias + LeakyReLU(0.2) in the
epilogue, 1-bit sign masks for the backward); the two classifier heads are one GEMM (cat(cls1, cls2) padded to 64 columns).
The three soft-label cross-entropies of an iteration never materialise a [B,C,H,W] tensor: soft labels are rebuilt per
pixel from the 1/8-resolution segmentation logits inside `mi_upsample_softce`.
"""
import datetime
import os
import time

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .. import kernels as K
from . import arch, ddp, engine
from .metrics import (MetricLogger, adjust_learning_rate, dump_json, setup_logger, soft_label_cross_entropy,  # noqa: F401
                      strip_prefix_if_present)
from .modules import _require_gpu
from .trainer import ASPPTrainer, _cpu_store

LEAK = 0.2
NPAD = 64          # merged classifier head: 2*num_classes real channels, padded so its data gradient has Cin % 64 == 0


class _DiscEngine:
    def __init__(self, owner):
        self.owner = owner
        self._sig = None
        self.packs = None
        self._have_dgrad = False

    def _params(self):
        o = self.owner
        return [o.D._modules["0"].weight, o.D._modules["0"].bias, o.D._modules["2"].weight, o.D._modules["2"].bias,
                o.cls1.weight, o.cls1.bias, o.cls2.weight, o.cls2.bias]

    def prepare(self, train):
        ps = self._params()
        store = getattr(ps[0], "_mi_store", None)
        sig = (store.generation if store is not None else -1, sum(p._version for p in ps), ps[0].data_ptr())
        if sig == self._sig and (self._have_dgrad or not train):
            return
        w1, b1, w2, b2, wc1, bc1, wc2, bc2 = [p.detach() for p in ps]
        k2 = wc1.shape[0] * 2
        wc = torch.zeros((NPAD,) + tuple(wc1.shape[1:]), dtype=torch.float32, device=w1.device)
        wc[:k2 // 2] = wc1
        wc[k2 // 2:k2] = wc2
        bc = torch.zeros(NPAD, dtype=torch.float32, device=w1.device)
        bc[:k2 // 2] = bc1
        bc[k2 // 2:k2] = bc2
        ones = lambda n: torch.ones(n, dtype=torch.float32, device=w1.device)
        self.packs = dict(
            w1=K.pack_weight_fwd(w1.contiguous()), w2=K.pack_weight_fwd(w2.contiguous()), wc=K.pack_weight_fwd(wc),
            b1=b1.contiguous(), b2=b2.contiguous(), bc=bc, o1=ones(w1.shape[0]), o2=ones(w2.shape[0]), oc=ones(NPAD))
        if train:
            self.packs.update(w1t=K.pack_weight_dgrad(w1.contiguous()), w2t=K.pack_weight_dgrad(w2.contiguous()), wct=K.pack_weight_dgrad(wc))
        self._sig, self._have_dgrad = sig, train

    def forward(self, x, save):
        """x [B,h,w,C] bf16 -> d_low [B,h,w,64] fp32 (first 2K channels = cat(cls1, cls2))."""
        P = self.packs
        B, h, w, _ = x.shape
        hw = (h, w)
        bits = lambda n: torch.empty((B, h, w, n // 16), dtype=torch.int16, device=x.device) if save else None
        m1, m2 = bits(P["o1"].numel()), bits(P["o2"].numel())
        a1 = K.conv_gemm(x, P["w1"], hw, 3, 1, 1, 1, scale=P["o1"], bias=P["b1"], relu=True, leaky=LEAK, mask_out=m1)
        a2 = K.conv_gemm(a1, P["w2"], hw, 3, 1, 1, 1, scale=P["o2"], bias=P["b2"], relu=True, leaky=LEAK, mask_out=m2)
        dlow = K.conv_gemm(a2, P["wc"], hw, 3, 1, 1, 1, scale=P["oc"], bias=P["bc"], out_f32=True)
        return dlow, ((x, a1, a2, m1, m2) if save else None)

    def backward(self, saved, ddlow, need_dx, need_wgrad):
        x, a1, a2, m1, m2 = saved
        P = self.packs
        ps = self._params()
        hw = (x.shape[1], x.shape[2])
        g = ddlow.to(torch.bfloat16).contiguous()
        k = ps[4].shape[0]
        side = engine._SideStream.get(x.device) if need_wgrad else None      # weight gradients beside the data-gradient chain

        def head_grads():
            dwc = torch.empty((NPAD,) + tuple(ps[4].shape[1:]), dtype=torch.float32, device=x.device)
            K.conv_wgrad(g, a2, dwc, 3, 1, 1, 1)
            dbc = torch.empty(NPAD, dtype=torch.float32, device=x.device)
            K.bias_grad_bf16(g, dbc)
            for p, val in ((ps[4], dwc[:k]), (ps[5], dbc[:k]), (ps[6], dwc[k:2 * k]), (ps[7], dbc[k:2 * k])):
                slot, acc = engine.grad_slot(p)
                slot.add_(val) if acc else slot.copy_(val)

        def conv_grads(dy, xin, pw, pb):
            slot, acc = engine.grad_slot(pw)
            K.conv_wgrad(dy, xin, slot, 3, 1, 1, 1, accumulate=acc)
            slot, acc = engine.grad_slot(pb)
            K.bias_grad_bf16(dy, slot, accumulate=acc)

        if need_wgrad:
            engine._off_path(side, head_grads, g, a2)
        ga2 = K.conv_gemm(g, P["wct"], hw, 3, 1, 1, 1, K.GATHER_DGRAD, bits=m2, leaky=LEAK)
        if need_wgrad:
            engine._off_path(side, lambda: conv_grads(ga2, a1, ps[2], ps[3]), ga2, a1)
        ga1 = K.conv_gemm(ga2, P["w2t"], hw, 3, 1, 1, 1, K.GATHER_DGRAD, bits=m1, leaky=LEAK)
        if need_wgrad:
            engine._off_path(side, lambda: conv_grads(ga1, x, ps[0], ps[1]), ga1, x)
        dx = K.conv_gemm(ga1, P["w1t"], hw, 3, 1, 1, 1, K.GATHER_DGRAD) if need_dx else None
        if side is not None:
            side.join()
        return dx


class _DiscFn(torch.autograd.Function):
    """feature [B,h,w,C] bf16 NHWC -> discriminator logits [B,h,w,64] fp32 NHWC (2K real channels)."""

    @staticmethod
    def forward(ctx, x, eng, *params):
        train = any(ctx.needs_input_grad)
        eng.prepare(train)
        dlow, saved = eng.forward(x, save=train)
        ctx.eng, ctx.saved = eng, saved
        return dlow

    @staticmethod
    def backward(ctx, ddlow):
        dx = ctx.eng.backward(ctx.saved, ddlow.contiguous().float(), ctx.needs_input_grad[0], any(ctx.needs_input_grad[2:]))
        ctx.saved = None
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class _DiscSoftLossFn(torch.autograd.Function):
    """weight * soft_label_cross_entropy(model_D(fea, size), cat(soft, 0) | cat(0, soft)) with soft = clip(softmax(up(seg)/T), .9):
    aspp_fada.py:96-121 fused; the gradient w.r.t. the low-resolution discriminator logits is produced in the same pass."""

    @staticmethod
    def forward(ctx, x, seg_low, eng, domain, size, weight, temperature, *params):
        train = any(ctx.needs_input_grad)
        eng.prepare(train)
        dlow, saved = eng.forward(x, save=train)
        loss_out, dd = K.upsample_softce(seg_low, dlow, size, domain, temperature, 0.9, want_grad=train, grad_scale=weight)
        ctx.eng, ctx.saved, ctx.dd = eng, saved, dd
        return loss_out[0] * weight

    @staticmethod
    def backward(ctx, gout):
        dx = ctx.eng.backward(ctx.saved, ctx.dd * gout, ctx.needs_input_grad[0], any(ctx.needs_input_grad[7:]))
        ctx.saved = ctx.dd = None
        return (dx,) + (None,) * (len(ctx.needs_input_grad) - 1)


class PixelDiscriminator(nn.Module):
    """discriminator.py:31-50.  state_dict keys D.0.{weight,bias}, D.2.{weight,bias}, cls1.*, cls2.*"""

    def __init__(self, input_nc, ndf=512, num_classes=1):
        super().__init__()
        if input_nc % 64 or ndf % 128 or 2 * num_classes > NPAD:
            raise NotImplementedError("PixelDiscriminator on the MI355X engine needs input_nc % 64 == 0, ndf % 128 == 0, num_classes <= 32")
        ref = nn.Conv2d                                    # only used for the default initialisation of each tensor
        self.D = arch.Holder()
        for idx, (o, c) in (("0", (ndf, input_nc)), ("2", (ndf // 2, ndf))):
            conv = ref(c, o, 3, 1, 1)
            node = arch.Holder()
            node.weight, node.bias = nn.Parameter(conv.weight.detach().clone()), nn.Parameter(conv.bias.detach().clone())
            self.D.add_module(idx, node)
        for name in ("cls1", "cls2"):
            conv = ref(ndf // 2, num_classes, 3, 1, 1)
            node = arch.Holder()
            node.weight, node.bias = nn.Parameter(conv.weight.detach().clone()), nn.Parameter(conv.bias.detach().clone())
            self.add_module(name, node)
        self.num_classes = num_classes
        self._engine = _DiscEngine(self)
        self._store = None

    def engine_parameters(self):
        return list(self.named_parameters())

    def ensure_flat(self):
        dev = self.cls1.weight.device
        if self._store is None or not self._store.intact() or self._store.data.device != dev:
            self._store = engine.FlatStore(self.engine_parameters(), dev)
        return self._store

    @staticmethod
    def _nhwc(x):
        x = x.permute(0, 2, 3, 1).contiguous()
        return x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)

    def forward(self, x, size=None):
        _require_gpu(x, "PixelDiscriminator")
        self.ensure_flat()
        dlow = _DiscFn.apply(self._nhwc(x), self._engine, *self._engine._params())[..., :2 * self.num_classes]
        if size is not None:
            return engine.UpsampleFn.apply(dlow.contiguous(), tuple(int(s) for s in size))
        return dlow.permute(0, 3, 1, 2)

    def soft_loss(self, x, seg_low, domain, size, weight=1.0, temperature=1.8):
        """Fused weight * soft_label_cross_entropy(self(x, size), soft labels of domain half `domain`); seg_low: the classifier's
        1/8-resolution logits [B,K,h,w] (detached) the soft labels are derived from."""
        _require_gpu(x, "PixelDiscriminator")
        self.ensure_flat()
        seg = seg_low.detach().permute(0, 2, 3, 1).contiguous().float()
        return _DiscSoftLossFn.apply(self._nhwc(x), seg, self._engine, int(domain), tuple(int(s) for s in size), float(weight),
                                     float(temperature), *self._engine._params())


def build_adversarial_discriminator(cfg, num_features=None, mid_nc=256):
    """core/models/build.py:33-53 (resnet branch)."""
    _, backbone_name = cfg.MODEL.NAME.split("_")
    if not backbone_name.startswith("resnet"):
        raise NotImplementedError("backbone %r: only the resnet family is on the MI355X hot path" % backbone_name)
    return PixelDiscriminator(2048 if num_features is None else num_features, mid_nc, num_classes=cfg.MODEL.NUM_CLASSES)


class FusedAdam(torch.optim.Adam):
    """torch.optim.Adam(betas, eps; no amsgrad / weight decay) with the update on a HIP kernel; torch's state_dict format.
    `grad_clamp` (attribute, default None): clamp every gradient to [-c, c] in place before the update - core/utils/utils.py:6-16
    clip_gradient(optimizer, 0.5) followed by optimizer.step() in pranet_trainer.py:59-60, as one launch per tensor."""
    grad_clamp = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("weight_decay", 0) != 0 or group.get("maximize"):
                raise NotImplementedError("FusedAdam implements the reference's configuration (fada_adapter.py:24)")
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                K.adam_step(p.data, g, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2, group["eps"], int(st["step"].item()),
                            grad_clamp=self.grad_clamp)
                if self.grad_clamp is not None and g is not p.grad:
                    p.grad.copy_(g)
                ps = getattr(p, "_mi_store", None)
                if ps is not None:
                    ps.generation += 1
        return loss


class FADAAdapter:
    """fada_adapter.py:6-31."""
    build_adversarial_discriminator = staticmethod(build_adversarial_discriminator)

    def __init__(self, cfg, tgt_train_loader, device):
        self.cfg = cfg
        self.device = device
        self.tgt_train_loader = tgt_train_loader
        self.start_adv_epoch = 1
        self.distributed = False
        self.init_params()

    def init_params(self):
        self.model_D = self.build_adversarial_discriminator(self.cfg)
        self.model_D.to(self.device)
        opt = FusedAdam if self.device.type == "cuda" else torch.optim.Adam
        self.optimizer_D = opt(self.model_D.parameters(), lr=self.cfg.SOLVER.BASE_LR_D, betas=(0.9, 0.99))
        self.reducer = None
        up = dist.is_available() and dist.is_initialized()
        self.distributed = up and dist.get_world_size() > 1
        if self.distributed or (up and os.environ.get("MI_DDP_FORCE") == "1"):
            store = self.model_D.ensure_flat() if hasattr(self.model_D, "ensure_flat") and self.device.type == "cuda" else _cpu_store(self.model_D)
            self.reducer = ddp.GradAllReducer([store], overlap=False)
            self.reducer.broadcast_parameters(0)

    def _load_checkpoint(self, checkpoint, logger):
        if "model_D" in checkpoint:
            logger.info("Loading model_D from {}".format(self.cfg.resume))
            self.model_D.load_state_dict(strip_prefix_if_present(checkpoint["model_D"], "module."))
        if "adv_epoch" in checkpoint:
            self.start_adv_epoch = checkpoint["adv_epoch"] + 1


class AsppFada:
    """aspp_fada.py:13-198: source segmentation loss + adversarial target loss for the generator, then two discriminator
    losses; checkpoints `AsppFada-{epoch}.pth`, `aspp_fada_chart_params.json`."""
    trainer_cls = ASPPTrainer
    adapter_cls = FADAAdapter
    TEMPERATURE = 1.8
    FUSED = True        # False: literal order of operations on materialised tensors even when the modules offer the fused entry points
    # Source and target crops through the backbone as ONE batch (round 5).  aspp_fada.py:80-104 runs the feature extractor twice, forward and backward,
    # on 2 x B/2 crops and lets the two backward passes accumulate into .grad; with FrozenBatchNorm every sample is processed independently, so one
    # forward over the concatenated batch, the two losses on its halves and ONE backward with d features = [d source | d target] give the same
    # gradients (the sums over pixels run in another order: fp32 rounding only).  Why: at 4 crops a conv launch fills 118 of 256 CUs - the tile
    # kernels take the same time for 4 crops as for 8.  Needs equal crop sizes and a frozen-BatchNorm backbone; False restores the two passes.
    BATCHED = True

    def __init__(self, name, cfg, src_train_loader, tgt_train_loader, local_rank):
        self.cfg = cfg
        self.logger = setup_logger(name + "_train", cfg.OUTPUT_DIR, local_rank)
        self.aspp = self.trainer_cls(name, cfg, src_train_loader, local_rank, self.logger)
        self.fada = self.adapter_cls(cfg, tgt_train_loader, self.aspp.device)
        if cfg.resume:
            self.fada._load_checkpoint(self.aspp.checkpoint, self.logger)
        self.lr_data, self.D_lr_data = [], []
        self.loss_seg_data, self.loss_adv_tgt_data, self.loss_D_src_data, self.loss_D_tgt_data = [], [], [], []
        self.iteration = 0

    def _save_checkpoint(self, adv_epoch, save_path):
        clone = lambda sd: {k: v.detach().clone() for k, v in sd.items()}
        torch.save({
            "adv_epoch": adv_epoch, "iteration": self.iteration,
            "feature_extractor": clone(self.aspp.feature_extractor.state_dict()), "classifier": clone(self.aspp.classifier.state_dict()),
            "optimizer_fea": self.aspp.optimizer_fea.state_dict(), "optimizer_cls": self.aspp.optimizer_cls.state_dict(),
            "model_D": clone(self.fada.model_D.state_dict()), "optimizer_D": self.fada.optimizer_D.state_dict()}, save_path)

    def _overlap(self, on):
        if getattr(self.aspp, "reducer", None) is not None:
            self.aspp.reducer.overlap = on

    @staticmethod
    def _reduce(part):
        """WORLD_SIZE > 1: average the gradients over ranks before the optimizer step (host/ddp.py)."""
        if getattr(part, "reducer", None) is not None:
            part.reducer.finish()

    # -- one iteration, aspp_fada.py:66-127 ------------------------------------------------------------------------------------
    def train_step(self, src_input, src_label, tgt_input, max_iter):
        a, f = self.aspp, self.fada
        a._throttle()                 # at most two iterations in flight (host/trainer.py RUN_AHEAD)
        self.iteration += 1
        lr = adjust_learning_rate(self.cfg.SOLVER.LR_METHOD, self.cfg.SOLVER.BASE_LR, self.iteration, max_iter, power=self.cfg.SOLVER.LR_POWER)
        lr_d = adjust_learning_rate(self.cfg.SOLVER.LR_METHOD, self.cfg.SOLVER.BASE_LR_D, self.iteration, max_iter, power=self.cfg.SOLVER.LR_POWER)
        for g in a.optimizer_fea.param_groups:
            g["lr"] = lr
        for g in a.optimizer_cls.param_groups:
            g["lr"] = lr * 10
        for g in f.optimizer_D.param_groups:
            g["lr"] = lr_d
        a.optimizer_fea.zero_grad()
        a.optimizer_cls.zero_grad()
        f.optimizer_D.zero_grad()
        dev = a.device
        src_input = src_input.to(dev, non_blocking=True)
        src_label = src_label.to(dev, non_blocking=True).long()
        tgt_input = tgt_input.to(dev, non_blocking=True)
        src_size, tgt_size = tuple(src_input.shape[-2:]), tuple(tgt_input.shape[-2:])
        T = self.TEMPERATURE
        fused = self.FUSED and hasattr(a.classifier, "loss") and hasattr(f.model_D, "soft_loss")
        batched = (fused and self.BATCHED and src_input.shape[1:] == tgt_input.shape[1:] and getattr(a.feature_extractor, "freeze_bn", False)
                   and dev.type == "cuda" and isinstance(f.model_D, PixelDiscriminator))
        if batched:
            ns = src_input.shape[0]
            fea = a.feature_extractor(torch.cat((src_input, tgt_input), 0))
            halves = fea.detach()
            src_fea = halves[:ns].requires_grad_(True)
            loss_seg = a.classifier.loss(src_fea, src_label, self.cfg.INPUT.IGNORE_LABEL, temperature=T)
            src_low = a.classifier.last_low
            self._overlap(True)                                 # every gradient a backward below touches is final when it returns
            loss_seg.backward()                                 # classifier gradients + d source features
            with torch.no_grad():
                tgt_low = a.classifier(halves[ns:])             # the classifier receives no gradient from the target pass
            # The discriminator runs forward ONCE, over both halves: aspp_fada.py evaluates model_D(tgt_fea) twice (:98 for the adversarial loss, :119
            # for its own) and model_D(src_fea) once, on the same features and the same weights - the logits are the same tensors.  Its backward for
            # the adversarial loss (data gradients only, target half) comes now; the one for its own two losses (weight gradients, both halves at
            # once: loss_D_src + loss_D_tgt accumulate in .grad either way) after the generator's update, as in the reference.
            D = f.model_D
            eng = D._engine
            D.ensure_flat()
            eng.prepare(True)
            dlow, saved = eng.forward(D._nhwc(halves), save=True)
            seg = lambda low: low.detach().permute(0, 2, 3, 1).contiguous().float()
            seg_s, seg_t = seg(src_low), seg(tgt_low)
            out_adv, dd_adv = K.upsample_softce(seg_t, dlow[ns:], tgt_size, 0, T, 0.9, want_grad=True, grad_scale=0.001)
            loss_adv_tgt = out_adv[0] * 0.001
            dx_tgt = eng.backward(tuple(t[ns:] for t in saved), dd_adv, True, False)
            dfea = torch.empty_like(fea)
            dfea[:ns].copy_(src_fea.grad)
            dfea[ns:].copy_(dx_tgt.permute(0, 3, 1, 2))
            fea.backward(dfea)                                  # the backbone's backward, once, over both halves
            self._reduce(a)
            a.optimizer_fea.step()
            a.optimizer_cls.step()
            f.optimizer_D.zero_grad()
            out_s, dd_s = K.upsample_softce(seg_s, dlow[:ns], src_size, 0, T, 0.9, want_grad=True, grad_scale=0.5)
            out_t, dd_t = K.upsample_softce(seg_t, dlow[ns:], tgt_size, 1, T, 0.9, want_grad=True, grad_scale=0.5)
            eng.backward(saved, torch.cat((dd_s, dd_t), 0), False, True)
            loss_D_src, loss_D_tgt = out_s[0] * 0.5, out_t[0] * 0.5
            self._reduce(f)
            f.optimizer_D.step()
        elif fused:
            src_fea = a.feature_extractor(src_input)
            loss_seg = a.classifier.loss(src_fea, src_label, self.cfg.INPUT.IGNORE_LABEL, temperature=T)
            src_low = a.classifier.last_low                     # 1/8-resolution logits (detached) -> soft labels
            self._overlap(False)                                # backbone gradients are final only after the target pass
            loss_seg.backward()
            self._overlap(True)
            tgt_fea = a.feature_extractor(tgt_input)
            with torch.no_grad():
                tgt_low = a.classifier(tgt_fea)                 # the classifier receives no gradient from the target pass
            d_params = list(f.model_D.parameters())
            for p in d_params:                                  # their gradients from this loss are zeroed before use (:114)
                p.requires_grad_(False)
            loss_adv_tgt = f.model_D.soft_loss(tgt_fea, tgt_low, 0, tgt_size, weight=0.001, temperature=T)
            loss_adv_tgt.backward()
            for p in d_params:
                p.requires_grad_(True)
            self._reduce(a)
            a.optimizer_fea.step()
            a.optimizer_cls.step()
            f.optimizer_D.zero_grad()
            loss_D_src = f.model_D.soft_loss(src_fea.detach(), src_low, 0, src_size, weight=0.5, temperature=T)
            loss_D_src.backward()
            loss_D_tgt = f.model_D.soft_loss(tgt_fea.detach(), tgt_low, 1, tgt_size, weight=0.5, temperature=T)
            loss_D_tgt.backward()
            self._reduce(f)
            f.optimizer_D.step()
        else:                                                   # literal order of operations for foreign modules
            src_fea = a.feature_extractor(src_input)
            src_pred = a.classifier(src_fea, src_size).div(T)
            loss_seg = F.cross_entropy(src_pred, src_label, ignore_index=self.cfg.INPUT.IGNORE_LABEL)
            self._overlap(False)
            loss_seg.backward()
            src_soft = F.softmax(src_pred, dim=1).detach()
            src_soft[src_soft > 0.9] = 0.9
            tgt_fea = a.feature_extractor(tgt_input)
            tgt_soft = F.softmax(a.classifier(tgt_fea, tgt_size).div(T), dim=1).detach()
            tgt_soft[tgt_soft > 0.9] = 0.9
            loss_adv_tgt = 0.001 * soft_label_cross_entropy(f.model_D(tgt_fea, tgt_size), torch.cat((tgt_soft, torch.zeros_like(tgt_soft)), 1))
            loss_adv_tgt.backward()
            self._reduce(a)
            a.optimizer_fea.step()
            a.optimizer_cls.step()
            f.optimizer_D.zero_grad()
            loss_D_src = 0.5 * soft_label_cross_entropy(f.model_D(src_fea.detach(), src_size), torch.cat((src_soft, torch.zeros_like(src_soft)), 1))
            loss_D_src.backward()
            loss_D_tgt = 0.5 * soft_label_cross_entropy(f.model_D(tgt_fea.detach(), tgt_size), torch.cat((torch.zeros_like(tgt_soft), tgt_soft), 1))
            loss_D_tgt.backward()
            self._reduce(f)
            f.optimizer_D.step()
        return dict(loss_seg=loss_seg.detach(), loss_adv_tgt=loss_adv_tgt.detach(), loss_D_src=loss_D_src.detach(),
                    loss_D_tgt=loss_D_tgt.detach(), lr=lr, lr_d=lr_d)

    def train(self):
        save_to_disk = self.aspp.local_rank == 0
        n_it = min(len(self.aspp.train_loader), len(self.fada.tgt_train_loader))
        self.iteration = (self.fada.start_adv_epoch - 1) * n_it
        max_iter = self.cfg.SOLVER.EPOCHS * n_it
        self.logger.info("#" * 20 + " Start Adversarial Training " + "#" * 20)
        meters = MetricLogger(delimiter="  ")
        self.aspp.feature_extractor.train()
        self.aspp.classifier.train()
        self.fada.model_D.train()
        start, end = time.time(), time.time()
        for epoch in range(self.fada.start_adv_epoch, self.cfg.SOLVER.EPOCHS + 1):
            for loader in (self.aspp.train_loader, self.fada.tgt_train_loader):      # DistributedSampler: new order per epoch
                sampler = getattr(loader, "sampler", None)
                if hasattr(sampler, "set_epoch"):
                    sampler.set_epoch(epoch)
            for (src_input, src_label, _), (tgt_input, _, _) in zip(self.aspp.train_loader, self.fada.tgt_train_loader):
                data_time = time.time() - end
                r = self.train_step(src_input, src_label, tgt_input, max_iter)
                vals = {k: float(r[k]) for k in ("loss_seg", "loss_adv_tgt", "loss_D_src", "loss_D_tgt")}
                meters.update(loss_seg=vals["loss_seg"], loss_adv_tgt=vals["loss_adv_tgt"], loss_D=vals["loss_D_src"] + vals["loss_D_tgt"],
                              loss_D_src=vals["loss_D_src"], loss_D_tgt=vals["loss_D_tgt"])
                meters.update(time=time.time() - end, data=data_time)
                end = time.time()
                self.lr_data.append(r["lr"])
                self.D_lr_data.append(r["lr_d"])
                self.loss_seg_data.append(vals["loss_seg"])
                self.loss_adv_tgt_data.append(vals["loss_adv_tgt"])
                self.loss_D_src_data.append(vals["loss_D_src"])
                self.loss_D_tgt_data.append(vals["loss_D_tgt"])
                if self.iteration % 20 == 0 or self.iteration == max_iter:
                    eta = str(datetime.timedelta(seconds=int(meters.time.global_avg * (max_iter - self.iteration))))
                    mem = torch.cuda.max_memory_allocated() / 1024.0 / 1024.0 if self.aspp.device.type == "cuda" else 0.0
                    self.logger.info(meters.delimiter.join(["Epoch: {epoch}", "eta: {eta}", "iter: {iter}", "{meters}", "lr: {lr:.6f}",
                                                            "max mem: {memory:.0f}"]).format(
                        epoch=epoch, eta=eta, iter=self.iteration, meters=str(meters), lr=r["lr"], memory=mem))
            if epoch % self.cfg.SOLVER.CHECKPOINT_PERIOD == 0 and save_to_disk:
                os.makedirs(self.cfg.OUTPUT_DIR, exist_ok=True)
                self._save_checkpoint(epoch, os.path.join(self.cfg.OUTPUT_DIR, "AsppFada-{}.pth".format(epoch)))
        total = time.time() - start
        self.logger.info("Total training time: {} ({:.4f} s / epoch)".format(str(datetime.timedelta(seconds=total)),
                                                                             total / max(self.cfg.SOLVER.EPOCHS, 1)))
        os.makedirs(self.cfg.OUTPUT_DIR, exist_ok=True)
        dump_json(os.path.join(self.cfg.OUTPUT_DIR, "aspp_fada_chart_params.json"), {
            "learning rate": self.lr_data, "discriminator learning rate": self.D_lr_data, "segmentation loss": self.loss_seg_data,
            "target adversarial loss": self.loss_adv_tgt_data, "source discriminator loss": self.loss_D_src_data,
            "target discriminator loss": self.loss_D_tgt_data})
