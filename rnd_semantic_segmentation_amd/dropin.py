"""Drop-in import surface: makes the reference's module paths resolve to this package's host-side mirror,
so the reference's own scripts / plugins (`from core.configs import cfg`,
`from core.trainers.aspp_trainer import ASPPTrainer`, `from base.base_trainer import BaseTrainer`, ...)
run on the MI355X engine unchanged.  Installed by the tiny top-level `core/` and `base/` packages.

Mapped: the DeepLabV2 hot path, its FADA adversarial step, the PraNet path (row N3) and the GALD / GCPA path (row N4); the
other model families of the reference (attn, vgg, the FADA combos of gald / attn) are out of scope and raise ImportError with that message.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys
import types

_PKG = "rnd_semantic_segmentation_amd.host."

ALIASES = {
    "core.configs": _PKG + "config",                                  # reference core/configs/__init__.py:1
    "core.configs.defaults": _PKG + "config",                         # core/configs/defaults.py (_C)
    "core.models.build": _PKG + "modules",                            # core/models/build.py:13-53
    "core.models.feature_extractor": _PKG + "modules",                # core/models/feature_extractor.py:34-52
    "core.models.classifiers.aspp.classifier": _PKG + "modules",      # core/models/classifiers/aspp/classifier.py
    "core.components.layers": _PKG + "modules",                       # core/components/layers.py:5-23
    "core.trainers.aspp_trainer": _PKG + "trainer",                   # core/trainers/aspp_trainer.py
    "core.models.discriminator": _PKG + "fada",                       # core/models/discriminator.py:31-50 (PixelDiscriminator)
    "core.adapters.fada_adapter": _PKG + "fada",                      # core/adapters/fada_adapter.py:6-31
    "core.combos.aspp_fada": _PKG + "fada",                           # core/combos/aspp_fada.py:13-198
    "core.testers.aspp_tester": _PKG + "tester",                      # core/testers/aspp_tester.py
    "core.utils.utility": _PKG + "metrics",                           # core/utils/utility.py (DeepLab subset)
    "core.utils.adapt_lr": _PKG + "metrics",                          # core/utils/adapt_lr.py:12-17
    "core.trainers.pranet_trainer": _PKG + "pranet",                  # core/trainers/pranet_trainer.py
    "core.testers.pranet_tester": _PKG + "pranet",                    # core/testers/pranet_tester.py
    "core.models.classifiers.pranet.PraNet_Res2Net": _PKG + "pranet", # PraNet, RFB_modified, aggregation
    "core.models.classifiers.pranet.Res2Net_v1b": _PKG + "pranet",    # Bottle2neck
    "core.utils.utils": _PKG + "pranet",                              # clip_gradient, AvgMeter (core/utils/utils.py:6-38)
    "core.trainers.gald_trainer": _PKG + "gald",                      # core/trainers/gald_trainer.py
    "core.models.classifiers.gcpacc.gcpa_cc2": _PKG + "gald",         # GCPAEncoder, GCPADecoder
    "core.testers.gald_tester": _PKG + "gald",                        # core/testers/gald_tester.py (made runnable: see GALDTester)
    "core.models.classifiers.gcpacc.gcpa_gald": _PKG + "gald",        # FAM (gcpa_gald.py:47-107)
    "core.models.classifiers.gcpacc.encoders.hardnet_68": _PKG + "gald",      # HarDBlock (hardnet_68.py:86-160)
    "core.models.classifiers.gcpacc.contextagg.ccnet": _PKG + "gald",         # CrissCrossAttention (ccnet.py:37-127)
    "core.models.classifiers.gcpacc.contextagg.GALDNet": _PKG + "gald",       # LocalAttenModule (GALDNet.py:124-157)
    "core.datasets.build": _PKG + "data",                             # core/datasets/build.py:5-30
    "base.base_trainer": _PKG + "plugin",                             # base/base_trainer.py
    "base.base_model": _PKG + "plugin",                               # base/base_model.py
}
PACKAGES = {"core.models", "core.models.classifiers", "core.models.classifiers.aspp", "core.models.classifiers.pranet", "core.models.classifiers.gcpacc", "core.models.classifiers.gcpacc.encoders", "core.models.classifiers.gcpacc.contextagg", "core.components", "core.trainers", "core.adapters", "core.combos",
            "core.testers", "core.utils", "core.datasets"}


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        if self.target is None:
            mod = types.ModuleType(spec.name)
            mod.__path__ = []
            return mod
        mod = importlib.import_module(self.target)
        if any(k.startswith(spec.name + ".") for k in ALIASES) and not hasattr(mod, "__path__"):
            mod.__path__ = []          # lets `core.configs.defaults` resolve under the aliased `core.configs`
        return mod

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path=None, target=None):
        if name in ALIASES:
            return importlib.machinery.ModuleSpec(name, _AliasLoader(ALIASES[name]))
        if name in PACKAGES:
            return importlib.machinery.ModuleSpec(name, _AliasLoader(None), is_package=True)
        if name.startswith(("core.", "base.")):
            raise ImportError("%s is not part of the MI355X hot path (SURVEY.md 8: DeepLabV2-R101+ASPP, FADA, PraNet, GALD are; the rest is out of scope)" % name)
        return None


def install():
    if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _AliasFinder())
