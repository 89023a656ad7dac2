"""CPU tests of the FADA host side: module surface, drop-in imports, and the AsppFada loop (with the
oracle's CPU modules substituted) against the oracle's literal restatement of aspp_fada.py:66-127."""
import json
import logging
import os

import numpy as np
import pytest
import torch

import _cases
from oracle import ref_model
from rnd_semantic_segmentation_amd.host import config as hc
from rnd_semantic_segmentation_amd.host import fada, metrics, synth

ROOT = os.path.dirname(os.path.dirname(__file__))


def adv_cfg(tmp_path, **over):
    c = hc.CfgNode(hc.default_tree())
    c.merge_from_file(os.path.join(ROOT, "configs", "deeplabv2_r101_adv.yaml"))
    opts = ["OUTPUT_DIR", str(tmp_path)]
    for k, v in over.items():
        opts += [k, v]
    c.merge_from_list(opts)
    return c


def test_discriminator_surface(golden_dir):
    D = fada.PixelDiscriminator(2048, 256, 19)
    ref = ref_model.RefPixelDiscriminator(2048, 256, 19)
    assert list(D.state_dict().keys()) == json.load(open(os.path.join(golden_dir, "g10_discriminator_keys.json")))
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    D.load_state_dict(ref.state_dict())                                       # a reference checkpoint's model_D loads
    with pytest.raises(RuntimeError, match="MI355X only"):
        D(torch.zeros(1, 2048, 5, 5))                                         # no CPU path behind the product module
    with pytest.raises(NotImplementedError):
        fada.PixelDiscriminator(100, 256, 19)
    c = adv_cfg("/tmp")
    assert c.SOLVER.BASE_LR_D == 1e-4 and isinstance(fada.build_adversarial_discriminator(c), fada.PixelDiscriminator)


def test_soft_label_cross_entropy_matches_oracle():
    g = torch.Generator().manual_seed(3)
    pred = torch.randn(2, 6, 5, 7, generator=g)
    soft = torch.softmax(torch.randn(2, 6, 5, 7, generator=g), 1)
    assert torch.allclose(metrics.soft_label_cross_entropy(pred, soft), ref_model.ref_soft_label_cross_entropy(pred, soft), rtol=1e-6)
    w = torch.rand(2, 5, 7, generator=g)
    want = torch.mean(w * torch.sum(-soft * torch.log_softmax(pred, 1), 1))
    assert torch.allclose(metrics.soft_label_cross_entropy(pred, soft, w), want, rtol=1e-6)


def test_dropin_fada_surface():
    from core.adapters.fada_adapter import FADAAdapter
    from core.combos.aspp_fada import AsppFada
    from core.models.build import build_adversarial_discriminator
    from core.models.discriminator import PixelDiscriminator
    from core.utils.utility import soft_label_cross_entropy
    assert AsppFada is fada.AsppFada and FADAAdapter is fada.FADAAdapter and PixelDiscriminator is fada.PixelDiscriminator
    assert callable(build_adversarial_discriminator) and soft_label_cross_entropy is metrics.soft_label_cross_entropy
    with pytest.raises(ImportError, match="hot path"):
        import core.combos.gald_fada  # noqa: F401


def _batches(n, seed0, with_label=True):
    out = []
    for i in range(n):
        x = torch.from_numpy(synth.synth_image(1, 33, 33, seed=seed0 + i))
        y = torch.from_numpy(synth.synth_label(1, 33, 33, 19, seed=seed0 + i))
        out.append((x, y, ["n%d" % i]))
    return out


def _cpu_combo(tmp_path, monkeypatch, epochs=2, resume=""):
    from rnd_semantic_segmentation_amd.host import trainer as tr
    cfg = adv_cfg(tmp_path, **{"SOLVER.EPOCHS": epochs, "resume": resume})
    cfg.freeze()

    def formula(m):
        synth.load_formula_weights(m)
        return m

    monkeypatch.setattr(tr.ASPPTrainer, "build_feature_extractor", staticmethod(lambda cfg: formula(ref_model.RefFeatureExtractor((1, 1, 1, 1)))))
    monkeypatch.setattr(tr.ASPPTrainer, "build_classifier", staticmethod(lambda cfg: formula(ref_model.RefASPP())))
    monkeypatch.setattr(fada.FADAAdapter, "build_adversarial_discriminator",
                        staticmethod(lambda cfg: formula(ref_model.RefPixelDiscriminator(2048, 256, 19))))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    monkeypatch.setattr(fada, "setup_logger", lambda *a, **k: logging.getLogger("test_fada_%s" % tmp_path.name))
    return fada.AsppFada("aspp_fada", cfg, _batches(2, 60), _batches(3, 80), 0), cfg


def test_aspp_fada_loop_matches_oracle_and_checkpoints(tmp_path, monkeypatch):
    combo, cfg = _cpu_combo(tmp_path, monkeypatch)
    assert combo.fada.start_adv_epoch == 1 and isinstance(combo.fada.optimizer_D, torch.optim.Adam)
    assert combo.fada.optimizer_D.defaults["betas"] == (0.9, 0.99)
    combo.train()
    assert combo.iteration == 4 and len(combo.loss_seg_data) == 4              # min(len(src), len(tgt)) iterations per epoch
    # the oracle's literal restatement on the same weights / inputs
    fe, cls, D = ref_model.RefFeatureExtractor((1, 1, 1, 1)), ref_model.RefASPP(), ref_model.RefPixelDiscriminator(2048, 256, 19)
    for m in (fe, cls, D):
        synth.load_formula_weights(m)
    of, oc = ref_model.make_optimizers(fe, cls, cfg.SOLVER.BASE_LR)
    od = torch.optim.Adam(D.parameters(), lr=cfg.SOLVER.BASE_LR_D, betas=(0.9, 0.99))
    it, want = 0, []
    for ep in range(2):
        for (xs, ys, _), (xt, _, _) in zip(_batches(2, 60), _batches(3, 80)):
            it += 1
            want.append(ref_model.ref_fada_step(fe, cls, D, of, oc, od, xs, ys, xt, it, 4, cfg.SOLVER.BASE_LR, cfg.SOLVER.BASE_LR_D))
    for key, got in (("loss_seg", combo.loss_seg_data), ("loss_adv_tgt", combo.loss_adv_tgt_data), ("loss_D_src", combo.loss_D_src_data),
                     ("loss_D_tgt", combo.loss_D_tgt_data), ("lr", combo.lr_data), ("lr_d", combo.D_lr_data)):
        assert np.allclose(got, [w[key] for w in want], rtol=1e-5, atol=1e-12), key
    for (k, p), (_, q) in zip(combo.fada.model_D.named_parameters(), D.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), k
    chart = json.load(open(tmp_path / "aspp_fada_chart_params.json"))
    assert set(chart) == {"learning rate", "discriminator learning rate", "segmentation loss", "target adversarial loss",
                          "source discriminator loss", "target discriminator loss"}
    ck = torch.load(tmp_path / "AsppFada-2.pth", map_location="cpu")
    assert set(ck) == {"adv_epoch", "iteration", "feature_extractor", "classifier", "optimizer_fea", "optimizer_cls", "model_D", "optimizer_D"}
    assert ck["adv_epoch"] == 2 and ck["iteration"] == 4 and "D.0.weight" in ck["model_D"]
    # resume (aspp_fada.py:19-20, fada_adapter.py:26-31): model_D and the adversarial epoch come back
    combo2, _ = _cpu_combo(tmp_path, monkeypatch, epochs=3, resume=str(tmp_path / "AsppFada-1.pth"))
    assert combo2.fada.start_adv_epoch == 2
    ck1 = torch.load(tmp_path / "AsppFada-1.pth", map_location="cpu")
    assert torch.equal(combo2.fada.model_D.state_dict()["cls2.bias"], ck1["model_D"]["cls2.bias"])
    combo2.train()
    assert combo2.iteration == 6
