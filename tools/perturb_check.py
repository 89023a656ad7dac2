"""Can the network-level parity tests fail?  Each case below breaks ONE thing in the engine's host code the way a wiring bug would (a conv's tap
offset, a harmonic link, a resize convention, the order of a concatenation, a missing residual), runs the named tests in a child pytest process on
the GPU box, and reports which of them went red.  The tests must fail for every perturbation and pass unperturbed (the `none` case).

  python tools/perturb_check.py [case ...]            -> gpurun_out/perturb_check.txt

The perturbations are applied by an environment variable read in tests/conftest.py-free fashion: this script writes a sitecustomize-style module
(gpurun_out/_perturb/mi_perturb.py) that monkeypatches rnd_semantic_segmentation_amd at import time, and puts it first on PYTHONPATH of the child."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    "none": "",
    # PraNet: the second 3x3 of every Res2Net group conv gets dilation 2 (same output shape: pad follows) - a tap offset
    "pranet_tap_offset": """
        from rnd_semantic_segmentation_amd.host import pranet
        _orig = pranet._res2net_units
        def units(*a, **k):
            us, blocks = _orig(*a, **k)
            u = blocks[5]["convs"][1]
            g = list(u.geom); g[4] = g[5] = 2; g[6] = g[7] = 2; u.geom = tuple(g)
            return us, blocks
        pranet._res2net_units = units
    """,
    # PraNet: the partial decoder's first upsampling uses align_corners=False
    "pranet_resize_convention": """
        from rnd_semantic_segmentation_amd.host import pranet
        _orig = pranet._Run.resize
        state = {"n": 0}
        def resize(self, x, factor, align, size=None):
            if factor == 2 and align:
                state["n"] += 1
                if state["n"] == 1:
                    align = False
            return _orig(self, x, factor, align, size=size)
        pranet._Run.resize = resize
    """,
    # PraNet: the residual of one bottleneck is dropped (out = relu(bn3(conv3(...))) without + x)
    "pranet_missing_residual": """
        from rnd_semantic_segmentation_amd.host import pranet
        _orig = pranet._Run.conv_bn
        def conv_bn(self, x, u, relu, add=None, out=None, out_f32=False):
            if u.key == "resnet.layer3.2.conv3":
                add = None
            return _orig(self, x, u, relu, add=add, out=out, out_f32=out_f32)
        pranet._Run.conv_bn = conv_bn
    """,
    # GALD: layer 6 of every 8+-layer HarDBlock reads layers (3, 4) instead of (5, 4) - same channel counts (14 / 16 / 20 / 40-wide), another tensor
    "gald_link_index": """
        from rnd_semantic_segmentation_amd.host import gald
        _orig = gald._hard_block_units
        def units(prefix, cin, growth, mul, n):
            layers, links, out_ch = _orig(prefix, cin, growth, mul, n)
            if n >= 8:
                assert links[5] == [5, 4] and layers[2].cout == layers[4].cout
                links[5] = [3, 4]
            return layers, links, out_ch
        gald._hard_block_units = units
    """,
    # GALD: the FAM concatenation swaps z2 and z3
    "gald_cat_order": """
        from rnd_semantic_segmentation_amd.host import gald
        _orig = gald._GaldRun.mulrelu
        def mulrelu(self, a, b, out=None):
            if out is not None and out.storage_offset() % (3 * 256) == 256:
                out = out.as_strided(out.size(), out.stride(), out.storage_offset() + 256)
            elif out is not None and out.storage_offset() % (3 * 256) == 512:
                out = out.as_strided(out.size(), out.stride(), out.storage_offset() - 256)
            return _orig(self, a, b, out=out)
        gald._GaldRun.mulrelu = mulrelu
    """,
    # GALD: the local attention gate is resized with align_corners=False instead of True (GALDNet.py:150)
    "gald_resize_convention": """
        from rnd_semantic_segmentation_amd.host import gald
        def la(run, x, units):
            g = run.dw_bn_relu(run.dw_bn_relu(x, units[0]), units[1])
            return run.gate(x, run.resize(g, None, False, size=(x.t.shape[1], x.t.shape[2])))
        gald._local_attention = la
    """,
}
TESTS = {"pranet": ["tests/test_gpu_pranet.py::test_pranet_teacher_forced_every_block_vs_oracle", "tests/test_gpu_pranet.py::test_pranet_whole_net_160_vs_reference_golden"],
         "gald": ["tests/test_gpu_gald.py::test_gald_teacher_forced_every_block_vs_oracle", "tests/test_gpu_gald.py::test_gald_whole_net_352_vs_reference_golden"]}


def main():
    cases = sys.argv[1:] or list(CASES)
    d = os.path.join(ROOT, "gpurun_out", "_perturb")
    os.makedirs(d, exist_ok=True)
    lines = []
    for name in cases:
        with open(os.path.join(d, "sitecustomize.py"), "w") as f:
            f.write("import sys\nsys.path.insert(0, %r)\n" % ROOT + textwrap.dedent(CASES[name]))
        fams = ["pranet", "gald"] if name == "none" else [name.split("_")[0]]
        for fam in fams:
            for t in TESTS[fam]:
                env = dict(os.environ, PYTHONPATH=d + os.pathsep + os.environ.get("PYTHONPATH", ""))
                r = subprocess.run([sys.executable, "-m", "pytest", t, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True)
                why = [l for l in r.stdout.splitlines() if l.startswith("E ")][:2]
                lines.append("%-26s %-64s %s   %s" % (name, t.split("::")[1], "PASSED" if r.returncode == 0 else "FAILED", " | ".join(w[:150] for w in why)))
                print(lines[-1], flush=True)
    open(os.path.join(ROOT, "gpurun_out", "perturb_check.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
