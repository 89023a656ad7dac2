#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats of bench.py (default two-stream schedule, and single-stream), then the
# four PMC passes (single stream: MI_WGRAD_STREAM=0 MI_BATCH_LANES=1, so that counters and durations are attributable per kernel).  Outputs under gpurun_out/<tag>_*.
# usage: bash tools/profile_round.sh <tag>
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-events"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_stats -o bench -- $B > $root/gpurun_out/${tag}_stats.log 2>&1 || echo "stats failed"
export MI_WGRAD_STREAM=0 MI_BATCH_LANES=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_stats_1s -o bench -- $B > $root/gpurun_out/${tag}_stats_1s.log 2>&1 || echo "stats 1s failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $root/gpurun_out/${tag}_pmcA -o bench -- $B > $root/gpurun_out/${tag}_pmcA.log 2>&1 || echo "pmcA failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/${tag}_pmcB -o bench -- $B > $root/gpurun_out/${tag}_pmcB.log 2>&1 || echo "pmcB failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/${tag}_pmcC -o bench -- $B > $root/gpurun_out/${tag}_pmcC.log 2>&1 || echo "pmcC failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $root/gpurun_out/${tag}_pmcD -o bench -- $B > $root/gpurun_out/${tag}_pmcD.log 2>&1 || echo "pmcD failed"
grep -h "^{" $root/gpurun_out/${tag}_stats.log $root/gpurun_out/${tag}_stats_1s.log | cut -c1-200
# device copies inside the loop vs set-up copies (the review's "150 copyBuffer launches per step")
python3 $root/tools/count_copies.py $root/gpurun_out/${tag}_stats/bench_kernel_trace.csv > $root/gpurun_out/${tag}_copies.txt 2>&1 || echo "count_copies failed"
cat $root/gpurun_out/${tag}_copies.txt
# the trace CSVs are large: keep only stats + counter collections
rm -f $root/gpurun_out/${tag}_*/bench_kernel_trace.csv
ls -la $root/gpurun_out/${tag}_*/ | head -40
# counters -> profiles/pmc.json (FETCH_SIZE x2 / WRITE_SIZE per MI355X_MICROARCH.md), stamped with the commit being profiled
python3 $root/profiles/make_pmc_json.py $root/gpurun_out/${tag}_pmcA $root/gpurun_out/${tag}_pmcB $root/gpurun_out/${tag}_pmcC $root/gpurun_out/${tag}_pmc.json $tag $root/gpurun_out/${tag}_pmcD > /dev/null 2>&1 || echo "pmc json failed"
