#!/bin/bash
# per-(kernel, grid) time table of a bench workload: bash tools/shape_trace.sh <workload> <steps> <warmup>  -> gpurun_out/shape_<workload>.txt
wl=${1:-pranet}; steps=${2:-4}; warm=${3:-2}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
MI_GRAPH=0 timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/shape_prof_$wl -o p -- python3 $root/bench.py --workload $wl --steps $steps --warmup $warm --no-cpu-baseline --no-kernel-events > $root/gpurun_out/shape_$wl.log 2>&1 || echo "trace of $wl failed"
f=$(find $root/gpurun_out/shape_prof_$wl -name 'p_kernel_trace.csv' | head -1)
python3 $root/tools/trace_by_shape.py $f $((steps + warm)) 70 > $root/gpurun_out/shape_$wl.txt
rm -rf $root/gpurun_out/shape_prof_$wl
