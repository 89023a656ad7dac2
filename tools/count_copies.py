"""How many device-to-device copies (`__amd_rocclr_copyBuffer`) does ONE training step issue?  Reads a rocprofv3 kernel trace (CSV) of a bench run and counts,
per step, the copy kernels that start between two consecutive optimizer launches (`--step-kernel`, `--per-step` of them per step), i.e. inside the loop -
the set-up copies (FlatStore construction, load_state_dict, buffer flattening) that a per-run average charges to the steps come before the first one.
usage: python tools/count_copies.py <kernel_trace.csv> [--step-kernel sgd_kernel] [--per-step 2]"""
import csv
import sys


def main():
    path = sys.argv[1]
    opts = dict(zip(sys.argv[2::2], sys.argv[3::2]))
    stepk, per = opts.get("--step-kernel", "sgd_kernel"), int(opts.get("--per-step", "2"))
    rows = sorted(((int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))), key=lambda t: t[0])
    marks = [i for i, (_, n) in enumerate(rows) if stepk in n]
    ends = marks[per - 1::per]                      # the last optimizer launch of every step
    before = sum(1 for _, n in rows[:ends[0] + 1] if "copyBuffer" in n) if ends else 0
    per_step = [sum(1 for _, n in rows[a + 1:b + 1] if "copyBuffer" in n) for a, b in zip(ends, ends[1:])]
    total = sum(1 for _, n in rows if "copyBuffer" in n)
    print("copyBuffer launches: %d in the run; %d before the end of the first step (set-up + step 1); per later step: %s" % (total, before, per_step))


if __name__ == "__main__":
    main()
