cd $GRAFT_REPO_ROOT
bash tools/profile_aux.sh r04 > gpurun_out/profile_aux_r04.log 2>&1
cd $GRAFT_REPO_ROOT
for w in pranet gald deeplab_bn; do python bench.py --workload $w > gpurun_out/r04_bench_$w.json 2> gpurun_out/r04_bench_$w.err; done
python bench.py > gpurun_out/r04_bench_builder.json 2> gpurun_out/r04_bench_builder.err
for f in gpurun_out/r04_bench_pranet.json gpurun_out/r04_bench_gald.json gpurun_out/r04_bench_deeplab_bn.json gpurun_out/r04_bench_builder.json; do cut -c1-260 $f; done
