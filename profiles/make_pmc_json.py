"""Derive profiles/pmc.json (per-kernel HBM traffic and MFMA busy per launch) from rocprofv3 counter CSVs.

Inputs are three separate rocprofv3 passes of the same bench command (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE
cannot share a pass; never combine --pmc with sys/hip/hsa traces):
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <A> -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc FETCH_SIZE  --output-format csv -d <B> -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE  --output-format csv -d <C> -- python3 bench.py ...
gfx950 corrections (guide, section HBM): FETCH_SIZE is in KiB and reports HALF of a wide coalesced stream -> x 1024 x 2;
WRITE_SIZE is in KiB and exact -> x 1024.
A fourth pass (optional, sixth argument) adds the L2 picture: TCC_REQ_sum / TCC_HIT_sum / TCC_MISS_sum per launch.
Usage: python profiles/make_pmc_json.py <A> <B> <C> [out.json [round [<D>]]]"""
import csv
import glob
import json
import sys
from collections import defaultdict

KERNELS = {"igemm_nt_kernel": "igemm_nt_kernel", "igemm_pp_kernel": "igemm_pp_kernel", "wgrad_tn_kernel": "wgrad_tn_kernel", "wgrad_q3_kernel": "wgrad_q3_kernel", "wgrad_s4_kernel": "wgrad_s4_kernel", "wgrad_tn256_kernel": "wgrad_tn256_kernel",
           "wgrad_reduce_multi_kernel": "wgrad_reduce_multi_kernel", "wgrad_reduce_kernel": "wgrad_reduce_kernel", "upce_pass1_kernel": "upce_pass1_kernel"}


def load(d):
    path = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    agg, n, dur = defaultdict(lambda: defaultdict(float)), defaultdict(int), defaultdict(float)
    for r in csv.DictReader(open(path)):
        k = next((v for key, v in KERNELS.items() if key in r["Kernel_Name"]), None)
        if k is None:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
        dur[(k, r["Counter_Name"])] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg, n, dur


def main():
    a, na, da = load(sys.argv[1])
    b, nb, _ = load(sys.argv[2])
    c, nc, _ = load(sys.argv[3])
    d4 = n4 = None
    if len(sys.argv) > 6 and glob.glob(sys.argv[6] + "/**/*_counter_collection.csv", recursive=True):
        d4, n4, _ = load(sys.argv[6])
    out = {}
    for k in KERNELS.values():
        L = na[(k, "GRBM_GUI_ACTIVE")]
        if not L:
            continue
        gui, mf = a[k]["GRBM_GUI_ACTIVE"], a[k]["SQ_VALU_MFMA_BUSY_CYCLES"]
        fetch = 2 * 1024 * b[k]["FETCH_SIZE"] / nb[(k, "FETCH_SIZE")]
        write = 1024 * c[k]["WRITE_SIZE"] / nc[(k, "WRITE_SIZE")]
        out[k] = {"launches": L, "avg_launch_us": round(da[(k, "GRBM_GUI_ACTIVE")] / L / 1e3, 2),
                  "mfma_busy_frac": round(mf / (gui / 8 * 1024), 4), "clock_ghz": round(gui / 8 / da[(k, "GRBM_GUI_ACTIVE")], 3),
                  "hbm_read_bytes_per_launch": round(fetch), "hbm_write_bytes_per_launch": round(write),
                  "hbm_bytes_per_launch": round(fetch + write)}
        if d4 is not None and n4[(k, "TCC_REQ_sum")]:
            req, hit = d4[k]["TCC_REQ_sum"] / n4[(k, "TCC_REQ_sum")], d4[k]["TCC_HIT_sum"] / n4[(k, "TCC_HIT_sum")]
            out[k].update({"l2_requests_per_launch": round(req), "l2_hit_frac": round(hit / req, 4) if req else None})
    import subprocess
    import os
    commit = None
    stamp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".commit_stamp")     # written before a gpurun call: the GPU box has no .git
    if os.path.exists(stamp):
        commit = open(stamp).read().strip()
    if not commit:
        try:
            commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
        except OSError:
            commit = None
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _digest import sources_sha
    out["_meta"] = {"sources_sha": sources_sha(), "commit": commit or None, "round": sys.argv[5] if len(sys.argv) > 5 else None,
                    "note": "per launch, averaged over every launch of the kernel in `bench.py --steps 5 --warmup 3` (single-stream schedule)"}
    dst = sys.argv[4] if len(sys.argv) > 4 else "profiles/pmc.json"
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
