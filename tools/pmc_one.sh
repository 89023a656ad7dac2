#!/bin/bash
# usage: tools/pmc_one.sh <tag> <script> [args...]   -> gpurun_out/pmc_<tag>_<pass>/  (one rocprofv3 --pmc pass per counter group)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for p in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES" \
         "GRBM_GUI_ACTIVE GRBM_COUNT" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr" \
         "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $p -d $root/gpurun_out/pmc_${tag}_$i -o out --output-format csv -- python3 $root/tools/"$@" > $root/gpurun_out/pmc_${tag}_$i.log 2>&1 || echo "pass $i failed"
done
ls $root/gpurun_out/pmc_${tag}_*/ 2>/dev/null | head
