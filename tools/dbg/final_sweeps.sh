# quick sweeps of the general conv's selection rules (bench.py --workload gald | pranet)
run() { echo -n "$1 $2: "; env $1 python bench.py --workload $2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for c in 32 64 96 128; do run MI_GCONV_BN_C=$c gald; done
for k in 1024 1536 2304 3072; do run MI_GCONV_KC32_WGS=$k gald; done
for k in 192 320 448; do run MI_GCONV_KS2_WGS=$k pranet; done
for k in 24 48 72; do run MI_GWM_STEPS=$k gald; done
