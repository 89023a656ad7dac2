"""Per-kernel-class hardware counters of ANY bench workload -> profiles/pmc_<workload>.json (the generic twin of make_pmc_json.py, which knows the
DeepLab kernels by name).

Inputs: the counter CSVs of four separate rocprofv3 passes of the same command (MI355X_MICROARCH.md, HBM / rocprofv3 section: FETCH_SIZE and
WRITE_SIZE cannot share a pass; --pmc is never combined with sys / hip / hsa traces):
    A: --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE     B: --pmc FETCH_SIZE     C: --pmc WRITE_SIZE     D (optional): --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
gfx950 corrections (same guide): FETCH_SIZE is in KiB and reports HALF of a wide coalesced stream -> x 1024 x 2; WRITE_SIZE is in KiB and exact.
A kernel class = the function name without template arguments / namespaces.  Steps in the trace = launches of `--step-kernel` (one per optimizer
step: adam, sgd ...) divided by `--per-step`.
Usage: python profiles/make_pmc_any.py <A> <B> <C> <out.json> <tag> [<D>] [--step-kernel NAME] [--per-step N]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

PEAK_HBM, PEAK_CLK_CYCLES = 8.0e12, 1024.0


def kclass(name):
    m = re.search(r"([A-Za-z_][A-Za-z_0-9]*_kernel)", name)
    if m:
        return m.group(1)
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.split(r"[<(]", name)[0].strip() or name[:40]


def load(d):
    paths = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
    if not paths:
        return None
    val, n, dur = defaultdict(float), defaultdict(int), defaultdict(float)
    for r in csv.DictReader(open(paths[0])):
        k, c = kclass(r["Kernel_Name"]), r["Counter_Name"]
        val[(k, c)] += float(r["Counter_Value"])
        n[(k, c)] += 1
        dur[(k, c)] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return val, n, dur


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = {}
    it = iter(sys.argv[1:])
    for a in it:
        if a.startswith("--"):
            opts[a[2:]] = next(it)
            args = [x for x in args if x != opts[a[2:]]]
    A, B, C = load(args[0]), load(args[1]), load(args[2])
    D = load(args[5]) if len(args) > 5 else None
    classes = sorted({k for (k, c) in A[1] if c == "GRBM_GUI_ACTIVE"})
    step_kernel, per_step = opts.get("step-kernel"), int(opts.get("per-step", "1"))
    steps = None
    if step_kernel:
        hits = [A[1][(k, "GRBM_GUI_ACTIVE")] for k in classes if step_kernel in k]
        steps = max(1, sum(hits) // per_step) if hits else None
    out, tot = {}, dict(us=0.0, rd=0.0, wr=0.0, launches=0)
    for k in classes:
        L = A[1][(k, "GRBM_GUI_ACTIVE")]
        gui, mf, t_ns = A[0][(k, "GRBM_GUI_ACTIVE")], A[0][(k, "SQ_VALU_MFMA_BUSY_CYCLES")], A[2][(k, "GRBM_GUI_ACTIVE")]
        rd = 2 * 1024 * B[0][(k, "FETCH_SIZE")] / max(B[1][(k, "FETCH_SIZE")], 1) if B else None
        wr = 1024 * C[0][(k, "WRITE_SIZE")] / max(C[1][(k, "WRITE_SIZE")], 1) if C else None
        e = {"launches": L, "avg_launch_us": round(t_ns / L / 1e3, 2), "mfma_busy_frac": round(mf / (gui / 8 * PEAK_CLK_CYCLES), 4) if gui else None,
             "hbm_read_bytes_per_launch": round(rd) if rd is not None else None, "hbm_write_bytes_per_launch": round(wr) if wr is not None else None}
        if rd is not None and wr is not None:
            e["hbm_bytes_per_launch"] = round(rd + wr)
            e["hbm_tb_s"] = round((rd + wr) / (t_ns / L) / 1e3, 3)
            e["hbm_frac_of_peak"] = round((rd + wr) / (t_ns / L) * 1e9 / PEAK_HBM, 4)
            tot["rd"] += rd * L
            tot["wr"] += wr * L
        if D and D[1][(k, "TCC_REQ_sum")]:
            req, hit = D[0][(k, "TCC_REQ_sum")] / D[1][(k, "TCC_REQ_sum")], D[0][(k, "TCC_HIT_sum")] / D[1][(k, "TCC_HIT_sum")]
            e.update({"l2_requests_per_launch": round(req), "l2_hit_frac": round(hit / req, 4) if req else None})
        if steps:
            e["launches_per_step"] = round(L / steps, 1)
            e["ms_per_step"] = round(t_ns / steps / 1e6, 3)
            if "hbm_bytes_per_launch" in e:
                e["hbm_gb_per_step"] = round(e["hbm_bytes_per_launch"] * L / steps / 1e9, 3)
        tot["us"] += t_ns / 1e3
        tot["launches"] += L
        out[k] = e
    commit = None
    stamp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".commit_stamp")
    if os.path.exists(stamp):
        commit = open(stamp).read().strip()
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _digest import sources_sha
    meta = {"sources_sha": sources_sha(), "commit": commit, "round": args[4] if len(args) > 4 else None, "steps_in_trace": steps,
            "note": "per launch, averaged over every launch of the kernel class in the profiled command (counter passes serialise the kernels)"}
    if steps:
        meta.update({"kernel_ms_per_step": round(tot["us"] / steps / 1e3, 3), "launches_per_step": round(tot["launches"] / steps, 1),
                     "hbm_read_gb_per_step": round(tot["rd"] / steps / 1e9, 3), "hbm_write_gb_per_step": round(tot["wr"] / steps / 1e9, 3),
                     "hbm_tb_s_over_kernel_time": round((tot["rd"] + tot["wr"]) / (tot["us"] * 1e-6) / 1e12, 3)})
    out["_meta"] = meta
    json.dump(out, open(args[3], "w"), indent=1)
    print(json.dumps(meta))


if __name__ == "__main__":
    main()
