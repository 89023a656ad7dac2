#!/bin/bash
# in-step A/B of the weight-gradient split pickers' slot counts (MI_WGRAD_S4_SLOTS: 1x1, MI_WGRAD_Q3_SLOTS: fused-row 3x3), interleaved rounds on one box
out=gpurun_out/r05_wgslots_ab2.txt; : > $out
for r in 1 2 3; do for cfg in "512 512" "512 448" "512 384" "512 320" "640 384" "768 512"; do set -- $cfg
  l=$(MI_WGRAD_S4_SLOTS=$1 MI_WGRAD_Q3_SLOTS=$2 python bench.py --no-cpu-baseline --no-kernel-events --steps 30 --warmup 8 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))")
  echo "round $r  S4_SLOTS=$1 Q3_SLOTS=$2  $l" | tee -a $out
done; done
