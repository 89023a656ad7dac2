"""rocprofv3 --kernel-trace CSV -> time per (kernel, grid) so that a kernel class can be split by the shapes it was launched on.
usage: python tools/trace_by_shape.py <p_kernel_trace.csv> [steps] [top]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
agg = defaultdict(lambda: [0, 0.0])
total = 0.0
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*$", "", name)
        m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)(I.*)?E[vP]", name)
        if m:
            name = m.group(1) + (" " + m.group(2)[:14] if m.group(2) else "")
        grid = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1), int(r["Grid_Size_Z"]))
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        k = (name[:60], grid)
        agg[k][0] += 1
        agg[k][1] += d
        total += d
print("total kernel time %.2f ms/step" % (total / steps * 1e-3))
print("%-62s %-18s %8s %8s %8s" % ("kernel", "workgroups", "n/step", "avg us", "ms/step"))
for (name, grid), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-62s %-18s %8.1f %8.1f %8.3f" % (name, "x".join(map(str, grid)), n / steps, t / n, t / steps * 1e-3))
