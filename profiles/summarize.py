"""Turn rocprofv3 CSV output (kernel_stats / counter_collection) into the markdown summaries kept in profiles/.
Usage: python profiles/summarize.py stats <kernel_stats.csv> <steps_in_trace> ; pmc <counter_collection.csv>"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[-72:]


def stats(path, steps):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("| kernel | calls/step | avg us | ms/step | % of GPU time |\n|---|---|---|---|---|")
    for r in rows[:28]:
        print("| `%s` | %.1f | %.1f | %.3f | %.1f |" % (short(r["Name"]), int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                    float(r["TotalDurationNs"]) / 1e6 / steps, 100 * float(r["TotalDurationNs"]) / tot))
    print("\nGPU kernel time per step: %.2f ms" % (tot / 1e6 / steps))


def pmc(path):
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(k, r["Counter_Name"])] += 1
    print("| kernel | counter | sum | per launch |\n|---|---|---|---|")
    for k in agg:
        for c, v in agg[k].items():
            print("| `%s` | %s | %.4g | %.4g |" % (k, c, v, v / calls[(k, c)]))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]))
    else:
        pmc(sys.argv[2])
