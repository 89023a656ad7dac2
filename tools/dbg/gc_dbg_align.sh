for d in 0 1 2 4 7; do echo "MI_GC_DBG=$d"; MI_GC_DBG=$d python tools/gkshape.py align fwd,dgrad 2>&1 | grep -v amdgpu | sed -n 2,5p; done
