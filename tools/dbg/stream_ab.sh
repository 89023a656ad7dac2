#!/bin/bash
# Interleaved A/B of the schedule switches on ONE box (boxes differ by 2-3 %): default (weight gradients on a side stream + two forward lanes), no side
# stream, no lanes, neither; R rounds each.  usage: bash tools/dbg/stream_ab.sh [rounds] [steps]
R=${1:-3}; S=${2:-30}
for r in $(seq 1 $R); do
  for cfg in "1 2" "0 2" "1 1" "0 1"; do
    set -- $cfg
    v=$(MI_WGRAD_STREAM=$1 MI_BATCH_LANES=$2 python bench.py --steps $S --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "round $r  MI_WGRAD_STREAM=$1 MI_BATCH_LANES=$2  $v"
  done
done
