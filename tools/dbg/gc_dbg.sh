for d in 0 1 2 4 6 7; do echo "MI_GC_DBG=$d"; MI_GC_DBG=$d python tools/gkshape.py pranet fwd 2>&1 | grep -v amdgpu | sed -n 2,9p; done
