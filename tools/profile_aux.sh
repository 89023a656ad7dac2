#!/bin/bash
# rocprofv3 kernel tables of the workloads beside the headline (the review's "driver-visible number for anything but config[1]"): PraNet, GALD,
# DeepLab with trainable BatchNorm (bench.py --workload ...), the FADA adversarial iteration and the fp32 evaluation path (their tools).
# usage: bash tools/profile_aux.sh <tag>      -> gpurun_out/<tag>_<name>_kernel_stats.csv, gpurun_out/<tag>_<name>.json / .log
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {     # name, program + args
    name=$1; shift
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_prof_$name -o p -- python3 "$@" > $root/gpurun_out/${tag}_$name.log 2>&1 || echo "$name failed"
    cp $root/gpurun_out/${tag}_prof_$name/p_kernel_stats.csv $root/gpurun_out/${tag}_${name}_kernel_stats.csv 2>/dev/null
    rm -rf $root/gpurun_out/${tag}_prof_$name
    grep -h "^{" $root/gpurun_out/${tag}_$name.log | tail -1 > $root/gpurun_out/${tag}_$name.json
    echo "$name: $(cut -c1-220 $root/gpurun_out/${tag}_$name.json)"
}
run pranet $root/bench.py --workload pranet --no-cpu-baseline --no-kernel-events
run gald $root/bench.py --workload gald --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events
run deeplab_bn $root/bench.py --workload deeplab_bn --steps 10 --warmup 4 --no-cpu-baseline --no-kernel-events
run fada $root/tools/fada_bench.py
run infer $root/tools/infer_bench.py
tail -3 $root/gpurun_out/${tag}_fada.log $root/gpurun_out/${tag}_infer.log
