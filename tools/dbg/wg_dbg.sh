for d in 0 1; do echo "MI_GW_DBG=$d"; MI_GW_DBG=$d python tools/gkshape.py pranet wgrad 2>&1 | grep -v amdgpu; done
