"""CPU tests of the host-side mirror: config node, LR, metrics vs the reference's golden outputs, state_dict keys,
drop-in import surface, trainer/tester plumbing (with the oracle's CPU model substituted), checkpoints."""
import json
import logging
import os

import numpy as np
import pytest
import torch

import _cases
from oracle import ref_model
from rnd_semantic_segmentation_amd.host import config as hc
from rnd_semantic_segmentation_amd.host import metrics, modules, synth


def fresh_cfg(tmp_path, **over):
    c = hc.CfgNode(hc.default_tree())
    c.merge_from_file(os.path.join(os.path.dirname(os.path.dirname(__file__)), "configs", "deeplabv2_r101_src.yaml"))
    opts = ["OUTPUT_DIR", str(tmp_path)]
    for k, v in over.items():
        opts += [k, v]
    c.merge_from_list(opts)
    return c


def test_cfgnode_semantics(tmp_path):
    c = hc.CfgNode(hc.default_tree())
    assert c.MODEL.NAME == "deeplab_resnet101" and c.SOLVER.BASE_LR == 0.02 and c.resume == ""
    y = tmp_path / "a.yaml"
    y.write_text("SOLVER:\n  BASE_LR: 5e-4\n  BATCH_SIZE: 6\nAUG:\n  COLLATE: None\nINPUT:\n  INPUT_SIZE_TEST: [2048, 1024]\n")
    c.merge_from_file(str(y))
    assert c.SOLVER.BASE_LR == 5e-4 and isinstance(c.SOLVER.BASE_LR, float)      # yacs literal_eval of '5e-4'
    assert c.AUG.COLLATE is None and c.INPUT.INPUT_SIZE_TEST == (2048, 1024)      # list -> tuple coercion
    c.merge_from_list(["SOLVER.EPOCHS", "3", "MODEL.FREEZE_BN", "True", "OUTPUT_DIR", "out"])
    assert c.SOLVER.EPOCHS == 3 and c.MODEL.FREEZE_BN is True and c.OUTPUT_DIR == "out"
    with pytest.raises(KeyError):
        c.merge_from_list(["SOLVER.NOPE", 1])
    with pytest.raises(ValueError):
        c.merge_from_list(["SOLVER.EPOCHS", "abc"])
    c.freeze()
    with pytest.raises(AttributeError):
        c.SOLVER.EPOCHS = 9
    d = c.clone()
    d.defrost()
    d.SOLVER.EPOCHS = 9
    assert c.SOLVER.EPOCHS == 3 and "BASE_LR: 0.0005" in str(c)


def test_metrics_match_reference_golden():
    g = _cases.load("g7_metrics")
    K = 19
    iu = metrics.intersectionAndUnionGPU(torch.from_numpy(g["pred"].copy()), torch.from_numpy(g["target"]), K)
    assert np.array_equal(np.stack([t.numpy() for t in iu]), g["iu"])
    iun = metrics.intersectionAndUnion(g["pred"].copy(), g["target2"], K)
    assert np.array_equal(np.stack(iun), g["iu2"])
    m = metrics.AverageMeter()
    m.update(*[a.astype(np.float64) for a in g["iu"]])
    m.update(*[a.astype(np.float64) for a in g["iu2"]])
    lines = []
    m.summary(type("L", (), {"info": lambda self, s: lines.append(s)})(), K)
    assert lines == [str(s) for s in g["summary"]]
    cfg = hc.CfgNode(hc.default_tree())
    cfg.MODEL.NUM_CLASSES = K
    assert np.array_equal(metrics.confusion_matrix(cfg, torch.from_numpy(g["small_p"]), torch.from_numpy(g["small_t"])).numpy(), g["cmt"])
    for it, lr in zip(g["lr_iters"], g["lrs"]):
        assert metrics.adjust_learning_rate("poly", 5e-4, int(it), 1000, 0.9) == pytest.approx(float(lr), rel=1e-15)
    with pytest.raises(NotImplementedError):
        metrics.adjust_learning_rate("step", 1, 1, 1, 1)
    sd = metrics.strip_prefix_if_present({"module.a": 1, "module.b": 2}, "module.")
    assert list(sd) == ["a", "b"] and list(metrics.strip_prefix_if_present({"a": 1, "module.b": 2}, "module.")) == ["a", "module.b"]


def test_product_state_dict_keys_equal_reference(golden_dir):
    full = json.load(open(os.path.join(golden_dir, "g8_r101_keys.json")))
    fe = modules.resnet_feature_extractor("resnet101", freeze_bn=True, pretrained_backbone=False)
    cls = modules.ASPP_Classifier_V2(2048, [6, 12, 18, 24], [6, 12, 18, 24], 19)
    assert list(fe.state_dict().keys()) + list(cls.state_dict().keys()) == full["keys"]
    assert sum(p.numel() for p in fe.parameters()) == full["n_fe_params"]
    assert sum(p.numel() for p in cls.parameters()) == full["n_cls_params"]
    tiny = json.load(open(os.path.join(golden_dir, "g8_tinynet_keys.json")))
    fe = modules.resnet_feature_extractor("resnet101", freeze_bn=True, pretrained_backbone=False, layers=(1, 1, 2, 2))
    assert list(fe.state_dict().keys()) + list(cls.state_dict().keys()) == tiny
    # checkpoints interchange with the oracle/reference layout
    ref = ref_model.RefFeatureExtractor((1, 1, 2, 2))
    synth.load_formula_weights(ref)
    fe.load_state_dict(ref.state_dict())
    assert torch.equal(fe.backbone.layer3._modules["1"].conv2.weight, ref.backbone.layer3._modules["1"].conv2.weight)


def test_product_refuses_cpu_and_unbuilt_variants():
    from rnd_semantic_segmentation_amd import _lib
    fe = modules.resnet_feature_extractor("resnet101", freeze_bn=True, pretrained_backbone=False, layers=(1, 1, 1, 1))
    with pytest.raises(_lib.MiError, match="MI355X only"):
        fe(torch.zeros(1, 3, 33, 33))
    cls = modules.ASPP_Classifier_V2(2048, [6, 12, 18, 24], [6, 12, 18, 24], 19)
    with pytest.raises(_lib.MiError, match="MI355X only"):
        cls(torch.zeros(1, 2048, 5, 5))
    bn = modules.resnet_feature_extractor("resnet101", freeze_bn=False, pretrained_backbone=False, layers=(1, 1, 1, 1))     # MODEL.FREEZE_BN=False
    assert isinstance(bn.backbone.bn1, torch.nn.BatchNorm2d) and "backbone.bn1.num_batches_tracked" in bn.state_dict()
    with pytest.raises(_lib.MiError, match="MI355X only"):
        bn(torch.zeros(2, 3, 33, 33))
    with pytest.raises(NotImplementedError):
        modules.ASPP_Classifier_V2(2048, [6, 12, 18, 24], [1, 1, 1, 1], 19)
    fe2 = modules.resnet_feature_extractor("resnet101", pretrained_weights="https://example.invalid/r101.pth", freeze_bn=True,
                                           layers=(1, 1, 1, 1))          # URL weights: skipped, never fetched
    assert isinstance(fe2, torch.nn.Module)


def test_dropin_import_surface():
    from base.base_model import BaseModel
    from base.base_trainer import BaseTrainer
    from core.components.layers import FrozenBatchNorm2d
    from core.configs import cfg
    from core.configs.defaults import _C
    from core.datasets.build import build_collate_fn, build_dataset
    from core.models.build import build_classifier, build_feature_extractor
    from core.models.classifiers.aspp.classifier import ASPP_Classifier_V2
    from core.models.feature_extractor import resnet_feature_extractor
    from core.testers.aspp_tester import ASPPTester
    from core.trainers.aspp_trainer import ASPPTrainer
    from core.utils.adapt_lr import adjust_learning_rate
    from core.utils.utility import AverageMeter, MetricLogger, inference, intersectionAndUnionGPU, setup_logger
    assert cfg is _C and issubclass(ASPPTrainer, BaseTrainer) and ASPP_Classifier_V2 is modules.ASPP_Classifier_V2
    with pytest.raises(ImportError, match="hot path"):
        import core.trainers.attn_trainer  # noqa: F401
    import core.trainers.gald_trainer as gt     # rows N3 / N4 of SURVEY 8f resolve (their classes need the GPU only when instantiated)
    import core.trainers.pranet_trainer as pt
    from core.models.classifiers.gcpacc.gcpa_cc2 import GCPADecoder, GCPAEncoder  # noqa: F401
    from core.models.classifiers.pranet.PraNet_Res2Net import PraNet  # noqa: F401
    from core.utils.utils import AvgMeter, clip_gradient  # noqa: F401
    assert issubclass(pt.PraNetTrainer, BaseTrainer) and issubclass(gt.GALDTrainer, BaseTrainer)
    from core.models.classifiers.gcpacc.contextagg.ccnet import CrissCrossAttention  # noqa: F401   (the GALD building blocks and the tester)
    from core.models.classifiers.gcpacc.contextagg.GALDNet import LocalAttenModule  # noqa: F401
    from core.models.classifiers.gcpacc.encoders.hardnet_68 import HarDBlock  # noqa: F401
    from core.models.classifiers.gcpacc.gcpa_gald import FAM  # noqa: F401
    from core.testers.gald_tester import GALDTester  # noqa: F401
    g = _cases.load("g4_frozenbn")
    bn = FrozenBatchNorm2d(96)
    bn.load_state_dict({k: torch.from_numpy(g[k]) for k in ("weight", "bias", "running_mean", "running_var")})
    assert np.abs(bn(torch.from_numpy(g["x"])).numpy() - g["y"]).max() < 1e-5

    class M(BaseModel):
        def forward(self, x):
            return x

    m = M({"a": 1})
    m.summary()


class _TinyLoader(list):
    pass


def _cpu_trainer(tmp_path, monkeypatch, epochs=2, n_batches=3, resume=""):
    """ASPPTrainer with the oracle's CPU modules substituted through its factory attributes."""
    from rnd_semantic_segmentation_amd.host import trainer as tr
    cfg = fresh_cfg(tmp_path, **{"SOLVER.EPOCHS": epochs, "SOLVER.BASE_LR": 5e-4, "resume": resume})
    cfg.freeze()

    def make_fe(cfg):
        m = ref_model.RefFeatureExtractor((1, 1, 1, 1))
        synth.load_formula_weights(m)
        return m

    def make_cls(cfg):
        m = ref_model.RefASPP()
        synth.load_formula_weights(m)
        return m

    monkeypatch.setattr(tr.ASPPTrainer, "build_feature_extractor", staticmethod(make_fe))
    monkeypatch.setattr(tr.ASPPTrainer, "build_classifier", staticmethod(make_cls))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    batches = _TinyLoader()
    for i in range(n_batches):
        x, lab = _cases.net_inputs(1, 33, 40 + i)
        batches.append((torch.from_numpy(x), torch.from_numpy(lab), ["n%d" % i]))
    log = logging.getLogger("test_trainer_%s" % tmp_path.name)
    return tr.ASPPTrainer("aspp", cfg, batches, 0, logger=log), cfg


def test_trainer_loop_lr_schedule_checkpoint_and_resume(tmp_path, monkeypatch):
    t, cfg = _cpu_trainer(tmp_path, monkeypatch)
    assert t.start_epoch == 1 and t.distributed is False and t.device.type == "cpu"
    t.train()
    assert t.iteration == 6 and len(t.loss_data) == 6 and len(t.lr_data) == 6
    want_lr = [metrics.adjust_learning_rate("poly", 5e-4, i, 6, 0.9) for i in range(6)]
    assert np.allclose(t.lr_data, want_lr, rtol=1e-12)                    # poly LR, incremented after the step
    assert t.optimizer_cls.param_groups[0]["lr"] == pytest.approx(10 * want_lr[-1])
    chart = json.load(open(tmp_path / "aspp_chart_params.json"))
    assert set(chart) == {"learning rate", "loss"} and len(chart["loss"]) == 6
    ck = torch.load(tmp_path / "Aspp-2.pth", map_location="cpu")
    assert set(ck) == {"epoch", "iteration", "feature_extractor", "classifier", "optimizer_fea", "optimizer_cls"}
    assert ck["epoch"] == 2 and ck["iteration"] == 6 and "backbone.conv1.weight" in ck["feature_extractor"]
    # the same steps done by the oracle's literal restatement of aspp_trainer.py:77-97 give the same losses
    fe, cls = ref_model.RefFeatureExtractor((1, 1, 1, 1)), ref_model.RefASPP()
    synth.load_formula_weights(fe)
    synth.load_formula_weights(cls)
    of, oc = ref_model.make_optimizers(fe, cls, 5e-4)
    it, ref_losses = 0, []
    for ep in range(2):
        for x, lab, _ in t.train_loader:
            loss, _ = ref_model.ref_train_step(fe, cls, of, oc, x, lab, it, 6, 5e-4)
            ref_losses.append(loss.item())
            it += 1
    assert np.allclose(t.loss_data, ref_losses, rtol=1e-5)
    # resume: start_epoch / iteration / optimizer state restored (aspp_trainer.py:28-44)
    t2, _ = _cpu_trainer(tmp_path, monkeypatch, epochs=3, resume=str(tmp_path / "Aspp-1.pth"))
    assert t2.start_epoch == 2 and t2.checkpoint["epoch"] == 1
    assert len(t2.optimizer_fea.state_dict()["state"]) > 0
    t2.train()
    assert t2.iteration == 9


def test_tester_loop_and_confusion_json(tmp_path, monkeypatch):
    from rnd_semantic_segmentation_amd.host import tester as te
    cfg = fresh_cfg(tmp_path)
    cfg.freeze()
    fe, cls = ref_model.RefFeatureExtractor((1, 1, 1, 1)), ref_model.RefASPP()
    synth.load_formula_weights(fe)
    synth.load_formula_weights(cls)
    # the stand-in classifier offers the engine's predict_probs() through the reference's own inference semantics
    cls.predict_probs = lambda feat, size: torch.softmax(
        torch.nn.functional.interpolate(cls(feat), size=size, mode="bilinear", align_corners=True), 1)
    monkeypatch.setattr(te.ASPPTester, "build_feature_extractor", staticmethod(lambda cfg: fe))
    monkeypatch.setattr(te.ASPPTester, "build_classifier", staticmethod(lambda cfg: cls))
    loader = []
    for i in range(2):
        x, lab = _cases.net_inputs(1, 33, 60 + i)
        loader.append((torch.from_numpy(x), torch.from_numpy(lab), ["t%d" % i]))
    names = {str(i): "c%d" % i for i in range(19)}
    lines = []
    logger = type("L", (), {"info": lambda self, s: lines.append(s), "warning": lambda self, s: None})()
    t = te.ASPPTester(cfg, torch.device("cpu"), loader, logger, [0] * 57, names)
    cmt = t.test()
    want = torch.zeros(19, 19, dtype=torch.int64)
    for x, lab, _ in loader:
        pred = ref_model.ref_inference(fe, cls, x, lab).max(1)[1]
        want += torch.from_numpy(ref_model_confusion(pred, lab))
    assert torch.equal(cmt, want)
    out = json.load(open(tmp_path / "aspp_confusion_matrix.json"))
    assert out["classes"] == list(names.values()) and np.array(out["cmt"]).sum() == int(want.sum())
    assert lines[0].startswith("Macro metric, val result: mIoU/mF1") and len(lines) == 2 + 2 * 19


def ref_model_confusion(pred, lab):
    from oracle import ref_ops
    return ref_ops.confusion_matrix(pred.numpy(), lab.numpy(), 19)


def test_synthetic_dataset_contract(tmp_path):
    from rnd_semantic_segmentation_amd.host import data
    cfg = fresh_cfg(tmp_path, **{"INPUT.SOURCE_INPUT_SIZE_TRAIN": "(96, 64)"})
    ds = data.build_dataset(cfg, "train", True)
    img, lab, name = ds[3]
    assert img.shape == (3, 64, 96) and img.dtype == torch.float32 and lab.shape == (64, 96) and lab.dtype == torch.float32
    vals = set(np.unique(lab.numpy()).tolist())
    assert vals <= set(range(19)) | {255.0} and 255.0 in vals and isinstance(name, str)
    assert abs(float(img.std()) - 1.0) < 0.1 and data.build_collate_fn(cfg) is None
    assert torch.equal(ds[3][0], img)                                         # deterministic


def test_optimizer_state_dict_interchanges_with_reference_order():
    """torch's optimizer state_dict is positional.  The reference builds torch.optim.SGD over
    module.parameters() (aspp_trainer.py:25-26): [w0, b0, w1, b1, ...] for the ASPP head.  The trainer here must use the same
    order so that `optimizer_cls` / `optimizer_fea` of a checkpoint load either way round with every momentum buffer on the
    parameter it belongs to."""
    from rnd_semantic_segmentation_amd.host import sgd, trainer
    rfe, rcls = ref_model.RefFeatureExtractor((1, 1, 1, 1)), ref_model.RefASPP()
    for m in (rfe, rcls):
        synth.load_formula_weights(m)
    ropt_f, ropt_c = ref_model.make_optimizers(rfe, rcls, 5e-4)
    for opt in (ropt_f, ropt_c):                       # one step with a recognisable gradient per parameter
        for gi, p in enumerate(opt.param_groups[0]["params"]):
            p.grad = torch.full_like(p, float(gi + 1))
        opt.step()
    fe = modules.resnet_feature_extractor("resnet101", freeze_bn=True, pretrained_backbone=False, layers=(1, 1, 1, 1))
    cls = modules.ASPP_Classifier_V2(2048, [6, 12, 18, 24], [6, 12, 18, 24], 19)
    for ref_mod, mod, ropt in ((rfe, fe, ropt_f), (rcls, cls, ropt_c)):
        assert [k for k, _ in mod.named_parameters()] == [k for k, _ in ref_mod.named_parameters()]
        params = trainer.ASPPTrainer._ordered_params(mod)
        assert all(a is b for a, b in zip(params, mod.parameters()))
        opt = sgd.FusedSGD(params, lr=1e-3, momentum=0.9, weight_decay=5e-4)
        opt.load_state_dict(ropt.state_dict())          # reference checkpoint -> this build
        for gi, p in enumerate(params):
            buf = opt.state[p]["momentum_buffer"]
            assert buf.shape == p.shape
            want = ropt.state[ropt.param_groups[0]["params"][gi]]["momentum_buffer"]
            assert torch.equal(buf, want)
        back = torch.optim.SGD(list(ref_mod.parameters()), lr=1e-3, momentum=0.9, weight_decay=5e-4)
        back.load_state_dict(opt.state_dict())          # this build's checkpoint -> the reference
        for p_ref, p in zip(ref_mod.parameters(), params):
            assert back.state[p_ref]["momentum_buffer"].shape == p_ref.shape


def test_pranet_warmup_cosine_schedule_equals_the_references_scheduler_chain():
    """warmup_cosine_lr against the learning rates the reference's own GradualWarmupScheduler(8, 5) + CosineAnnealingLR(100) chain produced for the
    first 40 epochs (g12_pranet_lr, pranet_trainer.py:97-104 / core/utils/adapt_lr.py:19-45): the linear warm-up AND the hand-over quirk (the
    cosine scheduler continues recursively from 8 * base_lr with its epoch counter already at 1: epoch 6 overshoots to 1.000247e-4)."""
    from rnd_semantic_segmentation_amd.host.pranet import warmup_cosine_lr
    g = _cases.load("g12_pranet_lr")
    base, lrs = float(g["base_lr"]), g["lrs"]
    assert len(lrs) == 40 and lrs[6] > lrs[5] == 8 * base                        # the overshoot is in the fixture
    for k, want in enumerate(lrs):
        assert abs(warmup_cosine_lr(base, k) / want - 1) < 1e-12, (k, warmup_cosine_lr(base, k), want)


def test_tile_route_of_the_tape_engines_is_consistent_with_the_padded_gathers():
    """host/pranet.py:_tile_route decides, from shapes alone, which convs of the tape engines (PraNet, GALD) go to the MFMA-tile kernels: 64-multiples on
    both sides as before, and (round 5) stride-1 layers whose channel counts pad to 32-multiples with < 1.6x the work, >= 16 384 pixels and >= 8 GFLOP -
    HarDNet-68's big gathered layers.  gald._hard_block pads a gather buffer exactly when this predicate holds, and _mfma_tile_ok takes the conv only when
    its input has the padded width: the two must agree or the general kernel would be handed a tensor wider than the conv's Cin."""
    from rnd_semantic_segmentation_amd.host import gald, pranet
    rup = pranet._rup32
    assert [rup(c) for c in (1, 32, 33, 466, 480)] == [32, 32, 64, 480, 480]
    layers, links, out_ch = gald._hard_block_units("b.", 256, 20, 1.7, 16)           # HarDNet-68's third block (hardnet_68.py:163-262: growth 20, 16 layers)
    assert out_ch == 328 and [(u.cin, u.cout) for u in layers][15] == (466, 168) and len(links[15]) == 5
    px = 6 * 90 * 160                                                                 # the block's map at 6 x 720 x 1280
    routed = [(u.cin, u.cout) for u, lk in zip(layers, links) if len(lk) > 1 and pranet._tile_route(u, px)]
    assert routed == [(310, 58), (368, 98), (466, 168)], routed                      # (152 -> 58 pads to 160 -> 64: only the 256-column loop could take it)
    # too few pixels, too little work, too much padding, strided, depthwise: the general kernel keeps them
    assert not pranet._tile_route(layers[15], 16383)
    assert not pranet._tile_route(pranet._Unit("c", "n", 96, 96, 3, 1, 1), 30976)                 # 5 GFLOP (PraNet's 96 -> 96 at 44 x 44 x 16)
    assert not pranet._tile_route(pranet._Unit("c", "n", 102, 40, 3, 1, 1), 345600)               # 2.0x the work when padded
    assert not pranet._tile_route(pranet._Unit("c", "n", 466, 168, 3, 2, 1), px)
    assert pranet._tile_route(pranet._Unit("c", "n", 256, 256, 3, 1, 1), 16384)                   # 64-multiples: as before, any stride
    assert pranet._tile_route(pranet._Unit("c", "n", 128, 64, 3, 2, 1), 16384)
    # the input the MFMA-tile kernels take for a padded layer is the padded gather buffer, nothing else
    u = layers[15]
    ok = lambda c: pranet._mfma_tile_ok(u, torch.empty((6, 90, 160, c), dtype=torch.bfloat16, device="meta"))
    assert ok(480) and not ok(466) and not ok(512)
