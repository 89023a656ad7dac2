"""profiles/<tag>_round_end_summary.md from the outputs of tools/profile_round.sh (kernel stats of the default and the single-stream schedule, pmc.json):
python profiles/make_round_summary.py r04 <steps in trace>"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[-72:]


def table(path, top=28):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = ["| kernel | calls/step | avg us | ms/step | % of GPU time |", "|---|---|---|---|---|"]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
        out.append("| `%s` | %.1f | %.1f | %.3f | %.1f |" % (short(r["Name"]), float(r["Calls"]) / steps, float(r["AverageNs"]) * 1e-3, float(r["TotalDurationNs"]) / steps * 1e-6,
                                                        100 * float(r["TotalDurationNs"]) / tot))
    out.append("\nGPU kernel time per step: %.2f ms" % (tot / steps * 1e-6))
    return "\n".join(out)


pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc.json")))
b = json.load(open(os.path.join(ROOT, "profiles", "%s_bench_builder.json" % tag)))
b100 = json.load(open(os.path.join(ROOT, "profiles", "%s_bench_100steps.json" % tag)))
fam = b["kernel_families"]
lines = ["# Round %s, end-of-round profile - 1x MI355X, B=8, 769x769, bf16 operands (commit %s)\n" % (tag[1:].lstrip("0"), pmc["_meta"].get("commit")),
         "Command (from /tmp on the GPU box, `tools/profile_round.sh %s`): `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-events`;" % tag,
         "%d training steps are in each trace (3 warm-up + 5 timed).  Raw tables: `%s_bench_kernel_stats.csv` (default two-stream schedule), `%s_bench_kernel_stats_single_stream.csv`." % (steps, tag, tag),
         "Bench line of the same commit (`%s_bench_builder.json`, default flags): **%.1f images/s, %.2f ms/step**; `igemm_pp_kernel` %.4f of the MFMA peak; dilated-3x3 family %.3f, ASPP head %.3f, "
         "K = 256 <-> 1024 class %.3f of HBM peak; CPU port %.3f images/s on %d cores.  `%s_bench_100steps.json` (`--steps 100 --warmup 20`, same box): sustained %.1f images/s, %.2f ms/step." % (
             tag, b["value"], b["ms_per_step"], b["roofline"]["frac"], fam["dilated3x3_family"]["frac_of_mfma_peak"], fam["aspp_head"]["frac_of_mfma_peak"],
             fam["pointwise_k256_n1024_class"]["frac_of_hbm_peak"], b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"], tag, b100["value"], b100["ms_per_step"]),
         "Driver, round 4 (`BENCH_r04.json`): 291.2 images/s, 27.475 ms, 0.4159 / 0.4197 / 0.3101; round 5 changed the schedule of the weight-gradient stream (slab reducers "
         "batched per block: +1.6 %% in an interleaved A/B on one box, `r05_reduce_batch_ab.txt`), not the conv kernels; boxes of the pool differ by 2 - 3 %% (this one: %.1f images/s; "
         "the A/B box: 300.0 with, 295.4 without)." % b["value"],
         "The other workloads and their counters: `%s_aux_summary.md`.\n" % tag,
         "## A. single-stream schedule (MI_WGRAD_STREAM=0 MI_BATCH_LANES=1): one kernel at a time, durations are the kernels' own\n",
         table(os.path.join(ROOT, "profiles", "%s_bench_kernel_stats_single_stream.csv" % tag)),
         "\n## B. default schedule (weight gradients on a second stream, two batch lanes): durations include co-scheduling\n",
         table(os.path.join(ROOT, "profiles", "%s_bench_kernel_stats.csv" % tag)),
         "\n## C. counters (single stream; `profiles/pmc.json`, FETCH_SIZE/WRITE_SIZE corrected per MI355X_MICROARCH.md)\n",
         "| kernel family | launches | avg us | MFMA busy | HBM MB/launch | L2 hit |", "|---|---|---|---|---|---|"]
total = 0.0
for k, v in pmc.items():
    if k == "_meta":
        continue
    lines.append("| `%s` | %d | %.1f | %.3f | %.1f | %s |" % (k, v["launches"], v["avg_launch_us"], v.get("mfma_busy_frac") or 0, v["hbm_bytes_per_launch"] / 1e6, v.get("l2_hit_frac")))
    total += v["launches"] * v["hbm_bytes_per_launch"]
lines.append("\nFabric bytes of these kernel families per step (%d steps in the trace): %.1f GB" % (steps, total / steps / 1e9))
open(os.path.join(ROOT, "profiles", "%s_round_end_summary.md" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:8]))
