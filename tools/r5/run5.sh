cd $GRAFT_REPO_ROOT
for v in evld; do echo "== $v"; SN_LIB=ab/$v.so python tools/r5/dbg_det.py 2>&1 | grep "^400\|vs run0" ; SN_LIB=ab/$v.so python tools/mask_profile.py mask 2>&1 | grep ms; done
