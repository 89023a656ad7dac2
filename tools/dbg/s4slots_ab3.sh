out=gpurun_out/r05_s4slots_ab3.txt; : > $out
for r in 1 2 3; do for v in 512 448 576; do
  l=$(MI_WGRAD_S4_SLOTS=$v python bench.py --no-cpu-baseline --no-kernel-events --steps 30 --warmup 8 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "round $r  S4_SLOTS=$v  $l" | tee -a $out
done; done
