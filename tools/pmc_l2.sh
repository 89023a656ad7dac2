#!/bin/bash
# usage: tools/pmc_l2.sh <tag> <script> [args...]  -> L2 hit / miss / fetch counters of one script (two rocprofv3 --pmc passes), summary on stdout
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for p in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $p -d $root/gpurun_out/l2_${tag}_$i -o out --output-format csv -- python3 $root/tools/"$@" > $root/gpurun_out/l2_${tag}_$i.log 2>&1 || echo "pass $i failed"
done
python3 - $root/gpurun_out/l2_${tag}_ <<'PY'
import csv, sys, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][-40:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "igemm" in k or "wgrad" in k:
        print(k, {c: "%.4g x%d" % (sum(v) / len(v), len(v)) for c, v in cs.items()})
PY
