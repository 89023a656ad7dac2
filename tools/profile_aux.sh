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
# hardware counters of a bench workload: four separate --pmc passes (the program straight after `--`, eager so that every launch is attributed),
# -> gpurun_out/<tag>_pmc_<name>.json (copy to profiles/pmc_<name>.json: bench.py reads it for roofline.traffic / the bound)
pmc() {     # name, step kernel, launches of it per step, bench args
    name=$1; stepk=$2; per=$3; shift 3
    i=0
    for ctr in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
        i=$((i+1))
        MI_GRAPH=0 timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $root/gpurun_out/${tag}_pmc${i}_$name -o p -- python3 $root/bench.py "$@" > $root/gpurun_out/${tag}_pmc${i}_$name.log 2>&1 || echo "pmc pass $i of $name failed"
        rm -f $root/gpurun_out/${tag}_pmc${i}_$name/*/p_kernel_trace.csv $root/gpurun_out/${tag}_pmc${i}_$name/p_kernel_trace.csv
    done
    python3 $root/profiles/make_pmc_any.py $root/gpurun_out/${tag}_pmc1_$name $root/gpurun_out/${tag}_pmc2_$name $root/gpurun_out/${tag}_pmc3_$name $root/gpurun_out/${tag}_pmc_$name.json $tag $root/gpurun_out/${tag}_pmc4_$name --step-kernel $stepk --per-step $per || echo "pmc json of $name failed"
    rm -rf $root/gpurun_out/${tag}_pmc?_$name
}
if [ "${PMC:-1}" = "1" ]; then
    pmc pranet adam 1 --workload pranet --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events
    pmc gald adam 2 --workload gald --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events
    pmc deeplab_bn sgd_kernel 2 --workload deeplab_bn --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events
    pmc fada adam 1 --workload fada --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events
fi
run pranet $root/bench.py --workload pranet --no-cpu-baseline --no-kernel-events
run gald $root/bench.py --workload gald --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events
run deeplab_bn $root/bench.py --workload deeplab_bn --steps 10 --warmup 4 --no-cpu-baseline --no-kernel-events
run fada $root/bench.py --workload fada --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events
run infer $root/tools/infer_bench.py
tail -n 3 $root/gpurun_out/${tag}_fada.log $root/gpurun_out/${tag}_infer.log
