cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for lib in "" ab/a4.so; do echo "== ${lib:-HEAD}"; SN_LIB=$lib python tools/mask_profile.py mask 2>&1 | grep ms; done; done
SN_LIB=ab/a4.so python tools/r5/dbg16.py 2>&1 | grep "^4000" | cut -c1-70
