cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
for v in 0 1; do
  MI_TILE_PAD=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r05_padprof$v -o p -- python3 $root/bench.py --workload gald --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events > $root/gpurun_out/r05_padprof$v.log 2>&1
  cp $root/gpurun_out/r05_padprof$v/p_kernel_stats.csv $root/gpurun_out/r05_padprof${v}_kernel_stats.csv
  rm -rf $root/gpurun_out/r05_padprof$v
done
