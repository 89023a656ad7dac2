#!/bin/bash
# In-launch second-level reductions (MI_INLAUNCH: weight-gradient slabs / column sums; MI_BN_INLAUNCH: the conv's BatchNorm finalize) on or off, interleaved on
# one box: PraNet (graph replay, the trainer's default, and eager) and GALD (eager).  usage: bash tools/dbg/inlaunch_ab.sh [rounds]
R=${1:-2}
for r in $(seq 1 $R); do
  for cfg in "1 1" "0 0" "1 0" "0 1"; do
    set -- $cfg
    for w in pranet gald; do
      v=$(MI_INLAUNCH=$1 MI_BN_INLAUNCH=$2 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], (d.get('eager') or {}).get('value'))")
      echo "round $r  MI_INLAUNCH=$1 MI_BN_INLAUNCH=$2  $w  $v"
    done
  done
done
