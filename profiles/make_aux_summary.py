"""profiles/<tag>_aux_summary.md from the outputs of tools/profile_aux.sh: bench lines (gpurun_out/<tag>_<name>.json), per-kernel-class counters
(profiles/pmc_<name>.json) of the workloads beside the headline.   python profiles/make_aux_summary.py r04"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
names = {"pranet": "PraNet, 16 x 352 x 352 training step (`bench.py --workload pranet`; BASELINE config[3])",
         "gald": "GALD, 6 x 720 x 1280 training step (`bench.py --workload gald`)",
         "deeplab_bn": "DeepLabV2-R101 with MODEL.FREEZE_BN False, 8 x 769 x 769 (`bench.py --workload deeplab_bn`)",
         "fada": "FADA adversarial iteration, 4 source + 4 target crops of 769 x 769 (`bench.py --workload fada`; BASELINE config[4] on one GPU)"}
out = ["# Round %s: the workloads beside the headline - bench lines and hardware counters (1x MI355X)\n" % tag[1:].lstrip("0"),
       "`bash tools/profile_aux.sh %s`: per workload four separate `rocprofv3 --kernel-trace --pmc ...` passes (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; FETCH_SIZE; WRITE_SIZE;\n"
       "TCC_REQ / HIT / MISS; eager, the program straight after `--`) -> `profiles/pmc_<workload>.json` (`profiles/make_pmc_any.py`: FETCH_SIZE x 2 x 1024, WRITE_SIZE x 1024 per\n"
       "MI355X_MICROARCH.md), then a `--kernel-trace --stats` run of the bench line.  Counter passes serialise the kernels: durations below are each kernel's own.\n" % tag]
for name, title in names.items():
    pj = os.path.join(ROOT, "profiles", "pmc_%s.json" % name)
    bj = os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (tag, name))
    if not os.path.exists(pj):
        continue
    pmc = json.load(open(pj))
    meta = pmc["_meta"]
    out.append("## %s\n" % title)
    if os.path.exists(bj):
        b = json.load(open(bj))
        extra = ""
        if "eager" in b:
            extra = "; eager %.1f images/s (%.2f ms)" % (b["eager"]["value"], b["eager"]["ms_per_step"])
        r = b.get("roofline", {})
        out.append("Bench line (`%s`): **%.1f images/s, %.2f ms/step**%s; dominant class `%s`: bound `%s`, MFMA (algorithmic) %s, MFMA busy %s, HBM %s of peak, %s launches/step.\n" % (
            os.path.basename(bj), b["value"], b["ms_per_step"], extra, r.get("kernel"), r.get("bound"), r.get("mfma_frac_algorithmic", r.get("frac")), r.get("mfma_busy_frac_counters"),
            r.get("hbm_frac_counters"), b.get("launches_per_step")))
    out.append("Counters (commit %s, %s steps in the trace): kernel time %.2f ms/step in %.0f launches; HBM read %.1f GB + write %.1f GB per step = %.2f TB/s over the kernel time (%.2f of the 8 TB/s peak).\n" % (
        meta.get("commit"), meta.get("steps_in_trace"), meta["kernel_ms_per_step"], meta["launches_per_step"], meta["hbm_read_gb_per_step"], meta["hbm_write_gb_per_step"],
        meta["hbm_tb_s_over_kernel_time"], meta["hbm_tb_s_over_kernel_time"] / 8.0))
    out.append("| kernel class | launches/step | avg us | ms/step | MFMA busy | HBM GB/step | HBM TB/s | of HBM peak | L2 hit |\n|---|---|---|---|---|---|---|---|---|")
    rows = sorted(((k, v) for k, v in pmc.items() if k != "_meta"), key=lambda kv: -kv[1].get("ms_per_step", 0))
    for k, v in rows[:18]:
        out.append("| `%s` | %.1f | %.1f | %.3f | %.3f | %.2f | %.2f | %.2f | %s |" % (k[-40:], v.get("launches_per_step", 0), v["avg_launch_us"], v.get("ms_per_step", 0), v.get("mfma_busy_frac") or 0,
                                                                               v.get("hbm_gb_per_step", 0), v.get("hbm_tb_s", 0), v.get("hbm_frac_of_peak", 0), v.get("l2_hit_frac")))
    out.append("")
open(os.path.join(ROOT, "profiles", "%s_aux_summary.md" % tag), "w").write("\n".join(out))
print("\n".join(out)[:1200])
