cd $GRAFT_REPO_ROOT
for lib in ab/exp_d64.so; do echo "== $lib"; SN_LIB=$lib python tools/r5/dbg16.py 2>&1 | grep "^4000" | cut -c1-60; SN_LIB=$lib SN_MASK16=8 python tools/mask_profile.py mask 2>&1 | grep ms; done
