#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5; mkdir -p $out
for lib in "" ab/early.so ab/w4.so; do echo "== ${lib:-HEAD}"; SN_LIB=$lib python tools/r5/dbg_det.py 2>&1 | grep "^400\|vs run0" | cut -c1-110; done > $out/run6_det.txt 2>&1; cat $out/run6_det.txt
for i in 1 2; do for lib in "" ab/early.so ab/w4.so ab/old3.so; do echo "== lib=${lib:-HEAD}"; SN_LIB=$lib timeout 300 python tools/mask_profile.py mask 2>&1 | grep ms; done; done > $out/run6_ab.txt 2>&1; cat $out/run6_ab.txt
SN_LIB=ab/wtrace.so timeout 300 python tools/mask_trace.py 2>&1 | grep -v amdgpu > $out/run6_trace.txt; cat $out/run6_trace.txt
