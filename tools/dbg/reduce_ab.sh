#!/bin/bash
# Slab reducers of a block's weight gradients batched into one launch (MI_WGRAD_REDUCE_BATCH=1) or one per conv (0): interleaved on one box.
R=${1:-3}; S=${2:-30}
for r in $(seq 1 $R); do
  for b in 1 0; do
    v=$(MI_WGRAD_REDUCE_BATCH=$b python bench.py --steps $S --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['final_loss'])")
    echo "round $r  MI_WGRAD_REDUCE_BATCH=$b  $v"
  done
done
