# per-shape time of the GALD convs by forced tile width
for f in 0 64 80 112 48; do echo "MI_GCONV_BN_FORCE=$f"; MI_GCONV_BN_ANY=1 MI_GCONV_BN_FORCE=$f python tools/gkshape.py gald fwd,dgrad 2>&1 | grep -v amdgpu | tail -n +2; done
