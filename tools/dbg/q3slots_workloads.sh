out=gpurun_out/r05_q3slots_workloads.txt; : > $out
for r in 1 2; do for v in 512 448; do for w in gald fada deeplab_bn; do
  l=$(MI_WGRAD_Q3_SLOTS=$v python bench.py --workload $w --no-cpu-baseline --no-kernel-events 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "round $r  Q3_SLOTS=$v $w  $l" | tee -a $out
done; done; done
