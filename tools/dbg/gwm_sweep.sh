# sweep of the batched weight-gradient launch: K steps per workgroup, fused kernel rows on / off
for st in 8 16 24 48; do for f in 1 0; do for w in pranet gald; do
  echo -n "steps=$st fused=$f $w: "; MI_GWM_STEPS=$st MI_GWM_FUSED3=$f python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done
