#!/bin/bash
# Experiment build of the library: the product sources compiled with -DMI_EXPERIMENTS, which adds the kernels that did not win and are
# therefore NOT part of libmi355seg.so (wgrad_p3_kernel, igemm_pw_kernel, the 256-wide tile of igemm_nt_kernel) and
# their switches (MI_WGRAD_P3, MI_IGEMM_PW, MI_IGEMM_BN=256, MI_P3_DBG).  Output: tools/experiments/libmi355seg_exp.so - point
# MI355SEG_LIB at it for tools/wgexp.py, tools/ppexp.py, tools/kexp.py.  Add -DMI_PP_TRACE for the timeline build of tools/pptrace.py.
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
csrc=$root/rnd_semantic_segmentation_amd/csrc
cd "$csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-result -DMI_EXPERIMENTS "$@" \
  -o "$root/tools/experiments/libmi355seg_exp.so" *.hip -ldl
echo "built $root/tools/experiments/libmi355seg_exp.so"
