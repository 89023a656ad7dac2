"""Headline benchmark (BASELINE.json): train images/s of DeepLabV2-ResNet101 + ASPP at 769x769, bf16 operands,
B = 8 per GPU, on N MI355X of one node (weak scaling), synthetic data resident in HBM, formula weights.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = reference core/trainers/aspp_trainer.py:77-95 (poly LR, zero_grad, forward, CE loss, backward, both SGD
steps) through host/trainer.py:ASPPTrainer.train_step on the HIP engine; N > 1 adds the RCCL gradient average.
Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the one with the largest total launch time in
the instrumented steps): algorithmic FLOPs / its launch time measured with HIP events on the launch stream inside the
timed region; `kernels` lists every timed kernel, `kernel_families` the north-star groups (dilated 3x3 family, ASPP head:
fraction of the MFMA peak; the K = 256 <-> N = 1024 pointwise class: fraction of the HBM peak).  `cpu_baseline` times the
oracle's torch-CPU port of the same step on all granted host cores (rank 0, N = 1).  MI_GRAPH=1 replays the step as a HIP
graph (no per-launch events then).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md chip table
INST = 3                       # instrumented steps (HIP events around every launch) run AFTER the timed region
PEAK_HBM_TBS = 8.0             # HBM3E peak, same table (6.29 TB/s measured with a float4 copy)
ALG_GFLOP_PER_IMAGE = 2518.5   # fwd + dgrad + wgrad of all 108 convs at 769x769


def synthetic_batch(batch, size, rank, device):
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x = torch.randn((batch, 3, size, size), generator=g)
    cell = 64                                                     # blocky class regions, like real label maps
    n = -(-size // cell)
    lab = torch.randint(0, 19, (batch, n, n), generator=g).repeat_interleave(cell, 1).repeat_interleave(cell, 2)[:, :size, :size].float().contiguous()
    band = 32 if size >= 256 else max(1, size // 24)
    lab[:, :band] = 255
    lab[:, -band:] = 255
    lab[:, :, :band] = 255
    lab[:, :, -band:] = 255
    lab[torch.rand((batch, size, size), generator=g) < 0.02] = 255
    return x.to(device), lab.to(device)


def head_commit():
    """The commit this tree is at: `git rev-parse` where .git exists, else the .commit_stamp file the profiling recipes write before a gpurun call (the
    GPU box receives the tree without .git), else None."""
    import subprocess
    try:
        r = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            return r.stdout.strip()
    except (OSError, subprocess.SubprocessError):
        pass
    try:
        return open(os.path.join(ROOT, ".commit_stamp")).read().strip() or None
    except OSError:
        return None


def counters_stale(meta):
    """True when the kernels / host schedule have changed since the hardware-counter file a roofline's `traffic` / HBM fraction come from was taken.
    `meta`: the file's `_meta`.  Compared by a digest of rnd_semantic_segmentation_amd/{csrc,host,*.py} (profiles/_digest.py; a docs-only commit does not
    make counters stale); files written before that digest existed fall back to the commit (None when either commit is unknown)."""
    meta = meta or {}
    if meta.get("sources_sha"):
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        from _digest import sources_sha
        return sources_sha() != meta["sources_sha"]
    head, profiled = head_commit(), meta.get("commit")
    if not head or not profiled:
        return None
    n = min(len(head), len(profiled))
    return head[:n] != profiled[:n]


def granted_cores():
    """Cores this process may actually use: the scheduler affinity capped by the cgroup CPU quota (a 1-GPU box exposes every
    core of the host in the affinity mask but grants a 16-core share; running 100+ threads on it takes minutes per step)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = int(txt[0]) / int(txt[1])
            else:
                q = int(txt[0])
                if q > 0:
                    quota = q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota is not None:
        n = min(n, max(1, int(quota)))
    else:
        n = min(n, 16)                                # no quota visible: the documented share of a 1-GPU box
    return max(1, n)


def cpu_baseline(size, budget_s=40.0):
    """The oracle's stock-PyTorch CPU restatement of the same training step (kind 'port'), B = 1, bounded sample."""
    from oracle import ref_model
    from rnd_semantic_segmentation_amd.host import synth
    cores = granted_cores()
    torch.set_num_threads(cores)                      # every core this process is granted
    fe, cls = ref_model.RefFeatureExtractor(), ref_model.RefASPP()
    synth.load_formula_weights(fe)
    synth.load_formula_weights(cls)
    of, oc = ref_model.make_optimizers(fe, cls, 5e-4)
    x, lab = synthetic_batch(1, size, 0, "cpu")
    ref_model.ref_train_step(fe, cls, of, oc, x, lab, 0, 100, 5e-4)      # warm-up (oneDNN primitive creation)
    n, t0 = 0, time.time()
    while n < 3 and (n == 0 or time.time() - t0 < budget_s):
        ref_model.ref_train_step(fe, cls, of, oc, x, lab, n + 1, 100, 5e-4)
        n += 1
    dt = time.time() - t0
    return {"value": round(n / dt, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d full train steps (fwd+CE+bwd+2xSGD) of B=1 %dx%d fp32 after 1 warm-up, oracle/ref_model.py" % (n, size, size)}



# ------------------------------------------------------------------------------------------------ the other two models of train_src.py
def aux_workload(args, device):
    """--workload pranet: BASELINE config[3] (configs/pranet_src_polyp.yaml: PraNet / Res2Net-50, batch 16, 352 x 352): one optimizer step = forward,
    four structure losses, backward, clamped Adam - the pass of the trainer's three-scale loop at rate 1 (pranet_trainer.py:44-61).
    --workload gald: configs/gald_src.yaml (HarDNet-68 encoder + GCPA decoder, batch 6, 1280 x 720): forward, four cross-entropies, backward, both Adams
    (gald_trainer.py).  Same JSON contract as the default workload: whole-step images/s, `roofline` for the dominant kernel of INST instrumented steps
    (HIP events around every launch), `cpu_baseline` = the oracle's torch-CPU restatement of the same step on the granted cores."""
    from rnd_semantic_segmentation_amd import kernels
    from rnd_semantic_segmentation_amd.host import gald, pranet, synth
    torch.manual_seed(0)
    if args.workload == "pranet":
        B, H, W = args.batch or 16, args.size or 352, args.size or 352
        net = pranet.PraNet().to(device).train()
        net.ensure_flat()
        opt = pranet.FlatAdam(net, 1e-4 / 8, grad_clamp=0.5)
        img, mask = synth.synth_polyp(B, H, W, seed=3)
        x, gt = torch.from_numpy(img).to(device), torch.from_numpy(mask).to(device)

        def step():
            opt.zero_grad()
            ls = [pranet.structure_loss(o, gt) for o in net(x)]
            (ls[3] + ls[2] + ls[1] + ls[0]).backward()
            opt.step()
            return ls[3]
        graphed = None
        if pranet.graph_mode_default():                   # PraNetTrainer's default: the whole optimizer step as one HIP graph (bit-equal to eager: tests/test_gpu_pranet.py); MI_GRAPH=0: eager
            graphed = pranet.GraphedStep(net, opt, x, gt)
        runner = (lambda: graphed()[3]) if graphed else step
        metric = "train images/sec at %dx%d bf16 (PraNet Res2Net-50, BASELINE config[3]; one step = ONE of the three passes of a reference iteration)" % (H, W)
        workload = ("configs/pranet_src_polyp.yaml: PraNet, one optimizer step = forward, four structure losses, backward, clamped Adam on B=%d %dx%d - the "
                    "reference's iteration (pranet_trainer.py:44-61) runs three such steps per batch (its three 'scales' all resize to trainsize)" % (B, H, W))
    elif args.workload == "fada":
        # BASELINE config[4] on one GPU: train_adv.py configs/deeplabv2_r101_adv.yaml - B/2 source + B/2 target crops through AsppFada.train_step
        # (reference core/combos/aspp_fada.py:80-127: source pass + CE at temperature 1.8, target pass + 0.001 x adversarial loss, both SGDs,
        # then the two 0.5 x discriminator losses on detached features and Adam; core/models/discriminator.py:31-50)
        import logging
        from rnd_semantic_segmentation_amd.host import config as hc, fada, modules
        from rnd_semantic_segmentation_amd.host import trainer as tr
        B, H, W = args.batch or 8, args.size or 769, args.size or 769
        cfg = hc.CfgNode(hc.default_tree())
        cfg.merge_from_file(os.path.join(ROOT, "configs", "deeplabv2_r101_adv.yaml"))
        cfg.merge_from_list(["OUTPUT_DIR", "/tmp/mi355seg_bench_fada"])
        cfg.freeze()

        def formula(m):
            synth.load_formula_weights(m)
            return m
        tr.ASPPTrainer.build_feature_extractor = staticmethod(lambda c: formula(modules.build_feature_extractor(c)))
        tr.ASPPTrainer.build_classifier = staticmethod(lambda c: formula(modules.build_classifier(c)))
        fada.FADAAdapter.build_adversarial_discriminator = staticmethod(lambda c: formula(fada.build_adversarial_discriminator(c)))
        fada.setup_logger = lambda *a, **k: logging.getLogger("bench_fada")
        combo = fada.AsppFada("aspp_fada", cfg, [], [], 0)
        hb = B // 2
        xs = torch.from_numpy(synth.synth_image(hb, H, W, seed=1)).to(device)
        ys = torch.from_numpy(synth.synth_label(hb, H, W, 19, seed=1)).to(device)
        xt = torch.from_numpy(synth.synth_image(hb, H, W, seed=2)).to(device)

        def step():
            return combo.train_step(xs, ys, xt, 10000)["loss_seg"]          # max_iter 10 000: the poly learning rate stays near its base value
        graphed, runner = None, step
        metric = "train images/sec at %dx%d bf16 (FADA adversarial iteration, source + target images; BASELINE config[4] on one GPU)" % (W, H)
        workload = ("train_adv.py configs/deeplabv2_r101_adv.yaml: one AsppFada iteration = %d source + %d target %dx%d crops, DeepLabV2-R101 + ASPP + "
                    "PixelDiscriminator, 2 x SGD + Adam" % (hb, hb, W, H))
    else:
        B, H, W = args.batch or 6, 720, 1280
        enc, dec = gald.GCPAEncoder().to(device).train(), gald.GCPADecoder().to(device).train()
        enc.ensure_flat()
        dec.ensure_flat()
        oe, od = pranet.FlatAdam(enc, 1e-4), pranet.FlatAdam(dec, 1e-3)
        x = torch.from_numpy(synth.synth_image(B, H, W, seed=9)).to(device)
        lab = torch.from_numpy(synth.synth_label(B, H, W, 19, seed=9)).to(device).long()

        def step():
            oe.zero_grad()
            od.zero_grad()
            l5, l4, l3, l2 = dec.losses(x, enc(x), lab)
            loss = l2 * 1 + l3 * 0.8 + l4 * 0.6 + l5 * 0.4
            loss.backward()
            oe.step()
            od.step()
            return loss
        graphed, runner = None, step
        metric = "train images/sec at %dx%d bf16 (GALD: HarDNet-68 + GCPA decoder)" % (W, H)
        workload = "configs/gald_src.yaml: GALD, one training step, B=%d %dx%d" % (B, W, H)
    note("warm-up")
    for _ in range(args.warmup):
        loss = runner()
    torch.cuda.synchronize()
    note("timed region")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = runner()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    out = {"metric": metric, "value": round(B / dt, 2), "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
           "data": "synthetic, random-init weights", "config": {"workload": workload, "hip_graph": bool(graphed)}, "loss": round(float(loss), 4)}
    if graphed:                                           # the same step eagerly (what MI_GRAPH=0 runs): both numbers in one line
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(3, args.steps // 2)):
            step()
        torch.cuda.synchronize()
        de = (time.perf_counter() - t0) / max(3, args.steps // 2)
        out["eager"] = {"value": round(B / de, 2), "ms_per_step": round(de * 1e3, 3)}
    if not args.no_kernel_events:
        note("instrumented steps (eager, HIP events per launch)")
        events = []
        kernels.PROFILE = events
        for _ in range(INST):
            step()
        kernels.PROFILE = None
        torch.cuda.synchronize()
        by = {}
        for name, e0, e1, flops, tag in events:
            v = by.setdefault(name, [0.0, 0.0, 0])
            v[0] += e0.elapsed_time(e1) * 1e-3
            v[1] += flops
            v[2] += 1
        tot = sum(v[0] for v in by.values())
        conv_fl = sum(v[1] for v in by.values())
        if os.environ.get("MI_BENCH_SHAPES"):          # per-shape table of the convs (stderr): where the conv time goes
            shapes = {}
            for name, e0, e1, flops, tag in events:
                if flops and tag is not None:
                    v = shapes.setdefault((name,) + tuple(tag), [0.0, 0.0, 0])
                    v[0] += e0.elapsed_time(e1) * 1e-3
                    v[1] += flops
                    v[2] += 1
            for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][0])[:40]:
                print("   %-70s %3d launches/step %8.1f us each %7.1f TFLOP/s %6.3f ms/step" % (k, v[2] // INST, 1e6 * v[0] / v[2], v[1] / v[0] / 1e12, 1e3 * v[0] / INST),
                      file=sys.stderr)
        dom = max(by, key=lambda k: by[k][0])
        tsec, fl, n = by[dom]
        # counters of the dominant kernel class from the separate rocprofv3 --pmc passes of tools/profile_aux.sh (profiles/pmc_<workload>.json,
        # written by profiles/make_pmc_any.py: FETCH_SIZE x 2 x 1024, WRITE_SIZE x 1024): they say which resource the class is nearer to
        pmc_class = {"gconv": "gconv_kernel", "gconv_wgrad": "gwgrad_kernel", "gconv_wgrad_multi": "gwgrad_multi_kernel", "gbn_bwd_sums": "gcolsum_partial_kernel", "gbn_bwd_apply": "gbn_bwd_apply_kernel",
                     "gbn_apply": "gbn_apply_kernel", "gbinary": "gbinary_kernel", "gbn_finalize": "gbn_finalize_kernel"}.get(dom, dom)
        traffic = hbm_frac = busy = src = whole = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_%s.json" % args.workload)))
            e = pmc[pmc_class]
            traffic, hbm_frac, busy = e.get("hbm_bytes_per_launch"), e.get("hbm_frac_of_peak"), e.get("mfma_busy_frac")
            src = {"file": "profiles/pmc_%s.json" % args.workload, "class": pmc_class, "profiled_commit": pmc["_meta"].get("commit"), "round": pmc["_meta"].get("round"),
                   "avg_launch_us_under_counters": e.get("avg_launch_us"), "l2_hit_frac": e.get("l2_hit_frac"), "counters_stale": counters_stale(pmc["_meta"])}
            whole = {k: pmc["_meta"].get(k) for k in ("kernel_ms_per_step", "launches_per_step", "hbm_read_gb_per_step", "hbm_write_gb_per_step", "hbm_tb_s_over_kernel_time")}
        except (OSError, KeyError, ValueError):
            pass
        mfma_frac = fl / tsec / 1e12 / PEAK_BF16_TFLOPS if fl else 0.0
        bound = "hbm" if (hbm_frac is not None and hbm_frac > max(mfma_frac, busy or 0.0)) or not fl else "mfma"
        alg_tbs = None
        if bound == "hbm":
            out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": round(hbm_frac * PEAK_HBM_TBS, 3) if hbm_frac is not None else None, "peak": PEAK_HBM_TBS, "unit": "TB/s",
                               "frac": hbm_frac, "traffic": traffic}
        else:
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": round(fl / tsec / 1e12, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(mfma_frac, 4),
                               "traffic": traffic}
        out["roofline"].update({"mfma_frac_algorithmic": round(mfma_frac, 4) if fl else None, "mfma_busy_frac_counters": busy, "hbm_frac_counters": hbm_frac, "traffic_source": src,
                                "launches_per_step": n // INST, "avg_launch_us": round(1e6 * tsec / n, 2), "ms_per_step_in_kernel": round(1e3 * tsec / INST, 3),
                                "note": "dominant = largest total launch time over %d eager instrumented steps after the timed region; algorithmic FLOPs = 2 * pixels * "
                                        "C_out * C_in * taps per launch (SURVEY 8d); `bound` = the resource whose counter fraction is larger for that kernel class "
                                        "(HBM bytes per launch / launch time against 8 TB/s, vs MFMA-busy cycles); with both fractions low the class is bound by "
                                        "neither: short contractions (26 .. 512 channels) whose K steps each cost one memory round trip" % INST})
        if whole:
            out["step_counters"] = whole
        out["whole_step_mfma_frac"] = round(conv_fl / INST / dt / 1e12 / PEAK_BF16_TFLOPS, 4)
        out["conv_gflop_per_step"] = round(conv_fl / INST / 1e9, 1)
        out["kernels"] = {k: {"ms_per_step": round(1e3 * v[0] / INST, 3), "launches_per_step": v[2] // INST, "share": round(v[0] / tot, 3),
                              "tflops": round(v[1] / v[0] / 1e12, 1) if v[1] else None} for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]}
        out["launches_per_step"] = sum(v[2] for v in by.values()) // INST
    if not args.no_cpu_baseline:
        note("CPU baseline (oracle port)")
        out["cpu_baseline"] = aux_cpu_baseline(args.workload, H, W)
    print(json.dumps(out), flush=True)


def aux_cpu_baseline(workload, H, W, budget_s=30.0):
    """The oracle's torch-CPU restatement of the same step (kind 'port'), bounded sample: B = 2 (BatchNorm needs a batch), fp32."""
    import torch.nn.functional as F
    from rnd_semantic_segmentation_amd.host import synth
    cores = granted_cores()
    torch.set_num_threads(cores)
    Bc = 2
    if workload == "fada":
        from oracle import ref_model
        fe, cls, D = ref_model.RefFeatureExtractor(), ref_model.RefASPP(), ref_model.RefPixelDiscriminator(2048, 256, 19)
        for m in (fe, cls, D):
            synth.load_formula_weights(m)
        of, oc = ref_model.make_optimizers(fe, cls, 2.5e-4)
        od = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.9, 0.99))
        xs = torch.from_numpy(synth.synth_image(1, H, W, seed=1)).float()
        ys = torch.from_numpy(synth.synth_label(1, H, W, 19, seed=1))
        xt = torch.from_numpy(synth.synth_image(1, H, W, seed=2)).float()
        cnt = [0]

        def step():
            cnt[0] += 1
            ref_model.ref_fada_step(fe, cls, D, of, oc, od, xs, ys, xt, cnt[0], 40000, 2.5e-4, 1e-4)
        what = "oracle/ref_model.py:ref_fada_step (1 source + 1 target image per step)"
    elif workload == "pranet":
        from oracle import ref_pranet
        net = ref_pranet.PraNet().train()
        opt = torch.optim.Adam(net.parameters(), 1e-4)
        img, mask = synth.synth_polyp(Bc, H, W, seed=3)
        x, gt = torch.from_numpy(img).float(), torch.from_numpy(mask).float()

        def step():
            opt.zero_grad()
            outs = net(x)
            sum(ref_pranet.structure_loss(o, gt) for o in outs).backward()
            for group in opt.param_groups:
                for p in group["params"]:
                    if p.grad is not None:
                        p.grad.data.clamp_(-0.5, 0.5)
            opt.step()
        what = "oracle/ref_pranet.py"
    else:
        from oracle import ref_gald
        enc, dec = ref_gald.GCPAEncoder().train(), ref_gald.GCPADecoder().train()
        oe, od = torch.optim.Adam(enc.parameters(), 1e-4), torch.optim.Adam(dec.parameters(), 1e-3)
        x = torch.from_numpy(synth.synth_image(Bc, H, W, seed=9)).float()
        lab = torch.from_numpy(synth.synth_label(Bc, H, W, 19, seed=9)).long()

        def step():
            oe.zero_grad()
            od.zero_grad()
            ref_gald.gald_losses(dec(x, enc(x)), lab)[1].backward()
            oe.step()
            od.step()
        what = "oracle/ref_gald.py"
    step()                                            # warm-up (oneDNN primitive creation)
    n, t0 = 0, time.time()
    while n < (2 if workload == "fada" else 3) and (n == 0 or time.time() - t0 < budget_s):
        step()
        n += 1
    dt = time.time() - t0
    return {"value": round(n * Bc / dt, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d full train steps of B=%d %dx%d fp32 after 1 warm-up, %s" % (n, Bc, W, H, what)}


_T0 = time.time()


def note(msg):
    print("[bench %7.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (BASELINE: 8; pranet 16, gald 6)")
    ap.add_argument("--size", type=int, default=None, help="crop side (BASELINE: 769; pranet 352)")
    ap.add_argument("--workload", choices=("deeplab", "deeplab_bn", "pranet", "gald", "fada"), default="deeplab",
                    help="deeplab = BASELINE config[1] (the headline, default); deeplab_bn = the same step with MODEL.FREEZE_BN False (trainable "
                         "BatchNorm2d on batch statistics); pranet = config[3]; gald = the third model of train_src.py; fada = config[4] on one GPU: train_adv.py's adversarial iteration, 4 source + 4 target crops")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py measures the MI355X path; no GPU is visible")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if args.workload not in ("deeplab", "deeplab_bn"):
        if world != 1:
            raise SystemExit("--workload %s is a one-GPU line" % args.workload)
        return aux_workload(args, device)
    args.batch, args.size = args.batch or 8, args.size or 769
    if world > 1 or os.environ.get("MI_DDP_FORCE") == "1":
        dist.init_process_group(backend="nccl", init_method="env://")       # RCCL
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from rnd_semantic_segmentation_amd import kernels
    from rnd_semantic_segmentation_amd.host import config as hc
    from rnd_semantic_segmentation_amd.host import synth
    from rnd_semantic_segmentation_amd.host.trainer import ASPPTrainer

    cfg = hc.CfgNode(hc.default_tree())
    cfg.merge_from_file(os.path.join(ROOT, "configs", "deeplabv2_r101_src.yaml"))
    cfg.merge_from_list(["OUTPUT_DIR", os.path.join(ROOT, "gpurun_out", "bench_out")])
    if args.workload == "deeplab_bn":
        cfg.merge_from_list(["MODEL.FREEZE_BN", "False"])
    cfg.freeze()
    import logging
    log = logging.getLogger("bench")
    log.addHandler(logging.NullHandler())
    note("building model")
    trainer = ASPPTrainer("aspp", cfg, [None] * 1000, local, logger=log)
    note("formula weights")
    with torch.no_grad():
        for m in (trainer.feature_extractor, trainer.classifier):
            synth.load_formula_weights(m)          # O(1) activations; random init explodes
            st = getattr(m, "_store", None)
            if st is not None:
                st.generation += 1
    x, lab = synthetic_batch(args.batch, args.size, rank, device)
    max_iter = 100000

    def step():
        loss, _ = trainer.train_step(x, lab, max_iter)
        trainer.iteration += 1
        return loss

    # Replay the step as a HIP graph (host/trainer.py: MI_GRAPH) where that is possible: one GPU, and enough warm-up steps for the
    # three eager steps + the capture to happen BEFORE the timed region.  The instrumented steps (per-launch HIP events) run eager.
    # Opt-in (MI_GRAPH=1): +3 % without per-launch events (269.6 vs 261.3 images/s on one box), within box-to-box noise with them.
    graph = os.environ.get("MI_GRAPH") == "1" and world == 1 and args.warmup >= 4 and os.environ.get("MI_DDP_FORCE") != "1"
    os.environ["MI_GRAPH"] = "1" if graph else "0"
    note("warm-up")
    for i in range(args.warmup):
        loss = step()
        if i == 0:
            torch.cuda.synchronize()
            note("first step done")
        if os.environ.get("MI_BENCH_VERBOSE"):
            note("warm-up step %d loss %.5f" % (i, float(loss)))
    torch.cuda.synchronize()
    note("timed region")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # The timed region is K steps of the PRODUCT schedule and nothing else (forward as two half-batch lanes, weight gradients beside the
    # data-gradient chain, bucketed exchange on its side stream).  The per-kernel numbers come from INST extra steps run after it, in
    # the same process: those keep every launch on the main stream with a HIP event pair around it (an event pair on a shared GPU
    # would charge each kernel its neighbours' time) and cost ~6 % per step - they are not part of `value`.
    red = getattr(trainer, "reducer", None)
    if red is not None:
        red.measure = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    note("timed region done: %.1f ms/step" % (1000 * elapsed / args.steps))
    exposed = red.exposed_ms() if red is not None else None
    if red is not None:
        red.measure = False
    events = [] if not args.no_kernel_events else None
    if events is not None:
        if graph:
            os.environ["MI_GRAPH"] = "0"          # events cannot be taken inside a graph replay
        kernels.PROFILE = events
        for i in range(INST):
            step()
        kernels.PROFILE = None
        torch.cuda.synchronize()
        note("instrumented steps done")
    final_loss = float(loss)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        images = args.batch * world * args.steps
        value = images / elapsed
        out = {
            "metric": "train images/sec at 769x769 bf16 (DeepLabV2-ResNet101 + ASPP)" + (", MODEL.FREEZE_BN False" if args.workload == "deeplab_bn" else ""),
            "value": round(value, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "train_src.py DeepLabV2-R101 bf16, %dx%d synthetic Cityscapes crops, batch %d per GPU, %dxMI355X%s"
                                   % (args.size, args.size, args.batch, world, ", trainable BatchNorm2d on batch statistics" if args.workload == "deeplab_bn" else ""),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "weights": "formula (synthetic)",
                       "hip_graph_replay": bool(graph),
                       "final_loss": round(final_loss, 5)},
            "whole_step_mfma_frac": round(ALG_GFLOP_PER_IMAGE * 1e9 * (args.size / 769.0) ** 2 * value / (PEAK_BF16_TFLOPS * 1e12 * world), 4),
        }
        if events:
            inst_steps = INST
            by, shapes = {}, {}
            fam = {"dilated3x3_family": [0.0, 0.0, 0], "aspp_head": [0.0, 0.0, 0], "pointwise_k256_n1024_class": [0.0, 0.0, 0]}
            for name, e0, e1, flops, tag in events:
                dt = e0.elapsed_time(e1) * 1e-3
                for d, key in ((shapes, tag), (by, name)):
                    v = d.setdefault(key, [0.0, 0.0, 0])
                    v[0] += dt
                    v[1] += flops
                    v[2] += 1
                kind, ksz, cin, cout, m = tag[0], tag[1], tag[2], tag[3], tag[4]
                dil = tag[6] if len(tag) > 6 else 1
                f = None
                if kind == "aspp_aux" or (kind == "fwd" and tag[5] & 32) or (kind in ("fwd", "dgrad") and cin == kernels.ASPP_KPAD) or (kind == "wgrad" and tag[5] == 1):
                    f = "aspp_head"                        # Z = X Wall^T, col2im | im2col, dX = G Wall, dWall = G^T X, bias sums
                elif ksz == 3 and dil > 1 and cin >= 256:
                    f = "dilated3x3_family"                # layer3 d2 (22 convs), layer4 d2 / d4 (3): forward, data and weight gradients
                elif kind in ("fwd", "dgrad") and ksz == 1 and cin == 256 and cout == 1024 and tag[5] in (71, 130):
                    f = "pointwise_k256_n1024_class"       # conv3 forward + residual, conv1 data gradient + residual: HBM-bound
                if f:
                    fam[f][0] += dt
                    fam[f][1] += (2.0 * m * (cin + 2 * cout) + 2.0 * m * cout / 8) if f.startswith("pointwise") else flops
                    fam[f][2] += 1
            dom = max(by, key=lambda k: by[k][0])
            tsec, fl, n = by[dom]
            # HBM bytes per launch of that kernel come from separate rocprofv3 --pmc passes (profiles/pmc.json, written by
            # profiles/make_pmc_json.py from the passes of tools/profile_round.sh): counters cannot be read inside this process.
            traffic, traffic_src, l2 = None, None, None
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc.json")))
                key = dom.split("+")[0]
                traffic = pmc[key]["hbm_bytes_per_launch"]
                if "l2_hit_frac" in pmc[key]:
                    l2 = {"hit_frac": pmc[key]["l2_hit_frac"], "requests_per_launch": pmc[key]["l2_requests_per_launch"],
                          "mfma_busy_frac": pmc[key]["mfma_busy_frac"]}
                traffic_src = {"file": "profiles/pmc.json", "profiled_commit": pmc.get("_meta", {}).get("commit"), "round": pmc.get("_meta", {}).get("round"),
                               "counters_stale": counters_stale(pmc.get("_meta", {}))}
            except (OSError, KeyError, ValueError):
                pass
            if dom == "bn_kernels":      # MODEL.FREEZE_BN False: the elementwise BatchNorm passes together outweigh any one conv kernel - HBM-bound
                passes = {0: 1, 1: 2, 2: 2, 3: 3, 4: 1}       # tensor passes of [M, C] bf16 per op (sums 1-2 reads, normalise read + write, input gradient 2 reads + write)
                nbytes = sum(passes[t[1]] * t[4] * t[2] * 2.0 for nm, _, _, _, t in events if nm == "bn_kernels")
                bn_traffic, bn_src = None, None
                try:                 # counters of the BatchNorm kernel classes (profiles/pmc_deeplab_bn.json, tools/profile_aux.sh): HBM bytes per launch, averaged over the classes
                    pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_deeplab_bn.json")))
                    cls = {k: v for k, v in pj.items() if k != "_meta" and ("bn_apply" in k or "bn_bwd_apply" in k or "bn_partial" in k)}
                    gb, ln = sum(v["hbm_gb_per_step"] for v in cls.values()), sum(v["launches_per_step"] for v in cls.values())
                    bn_traffic = round(gb * 1e9 / ln)
                    bn_src = {"file": "profiles/pmc_deeplab_bn.json", "classes": sorted(cls), "hbm_gb_per_step": round(gb, 2), "profiled_commit": pj["_meta"].get("commit"), "counters_stale": counters_stale(pj["_meta"]),
                              "hbm_frac_of_peak": {k: v.get("hbm_frac_of_peak") for k, v in cls.items()}}
                except (OSError, KeyError, ValueError, ZeroDivisionError):
                    pass
                out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": round(nbytes / tsec / 1e12, 3), "peak": PEAK_HBM_TBS, "unit": "TB/s",
                                   "frac": round(nbytes / tsec / 1e12 / PEAK_HBM_TBS, 4), "traffic": bn_traffic, "traffic_source": bn_src, "launches_per_step": n // inst_steps,
                                   "avg_launch_us": round(1e6 * tsec / n, 2), "ms_per_step_in_kernel": round(1e3 * tsec / inst_steps, 3),
                                   "alg_gbytes_per_step": round(nbytes / inst_steps / 1e9, 2),
                                   "note": "the normalise / backward-sums / input-gradient passes of trainable BatchNorm2d (csrc/batchnorm.hip): algorithmic bytes = "
                                           "tensor passes x M x C x 2 (residual reads and sign bits not counted)"}
            else:
              out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": round(fl / tsec / 1e12, 2), "peak": PEAK_BF16_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(fl / tsec / 1e12 / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src, "l2": l2,
                               "launches_per_step": n // inst_steps, "avg_launch_us": round(1e6 * tsec / n, 2),
                               "alg_gflop_per_launch": round(fl / n / 1e9, 3), "ms_per_step_in_kernel": round(1e3 * tsec / inst_steps, 3),
                               "instrumented_steps": inst_steps,
                               "note": "dominant = largest total launch time in the instrumented steps: %d extra steps AFTER the timed region, HIP events "
                                       "on the launch stream around every launch, single-stream so that an event pair times one kernel alone; the "
                                       "timed steps run the product schedule (two half-batch forward lanes, weight gradients beside the "
                                       "data-gradient chain on extra HIP streams) and carry no events" % INST}
            out["kernels"] = {k: {"ms_per_step": round(1e3 * v[0] / inst_steps, 3), "tflops": round(v[1] / v[0] / 1e12, 2) if v[1] else None,
                                  "frac_of_mfma_peak": round(v[1] / v[0] / 1e12 / PEAK_BF16_TFLOPS, 4) if v[1] else None,
                                  "launches_per_step": v[2] // inst_steps} for k, v in by.items()}
            families = {}
            for k, (tsec_f, work, n_f) in fam.items():
                if not n_f:
                    continue
                if k.startswith("pointwise"):
                    families[k] = {"bound": "hbm", "ms_per_step": round(1e3 * tsec_f / inst_steps, 3), "launches_per_step": n_f // inst_steps,
                                   "achieved_tb_s": round(work / tsec_f / 1e12, 3), "peak_tb_s": PEAK_HBM_TBS,
                                   "frac_of_hbm_peak": round(work / tsec_f / 1e12 / PEAK_HBM_TBS, 4),
                                   "alg_bytes": "per launch: bf16 A + residual + output, sign bits in / out"}
                else:
                    families[k] = {"bound": "mfma", "ms_per_step": round(1e3 * tsec_f / inst_steps, 3), "launches_per_step": n_f // inst_steps,
                                   "tflops": round(work / tsec_f / 1e12, 2), "frac_of_mfma_peak": round(work / tsec_f / 1e12 / PEAK_BF16_TFLOPS, 4)}
            out["kernel_families"] = families
        if events and os.environ.get("MI_BENCH_SHAPES"):
            note("in-step per-shape table: kind k Cin N M flags dil | launches/step  us/launch  TFLOP/s  ms/step")
            for tag, (tsec, fl, n) in sorted(shapes.items(), key=lambda kv: -kv[1][0]):
                note("%-8s k%d Cin%-5d N%-5d M%-7d f%-4d d%-2d | %3d  %8.1f  %7.0f  %6.3f" % (tuple(tag[:7]) + (n // inst_steps, 1e6 * tsec / n, fl / tsec / 1e12, 1e3 * tsec / inst_steps)))
        if red is not None:
            # data-parallel diagnostics of THIS rank (rank 0): how many ranks the process group really has, what travels, and how long the
            # compute stream waited for the exchange at the join before the optimizer (the part of the all-reduce that backward did not hide)
            out["ddp"] = dict(red.diag, exchange_ms_exposed=None if exposed is None else round(exposed, 4), backend=dist.get_backend() if dist.is_initialized() else None)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.size)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
