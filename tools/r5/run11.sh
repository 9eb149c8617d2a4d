#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_ops.py tests/test_gpu_render.py -m gpu -q -k "mask or head" 2>&1 | tail -3
export SN_LIB=sanerf-hq_amd/libsanerf_hip_exp.so
timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -k "mask16" 2>&1 | tail -2
for i in 1 2 3; do for m in 0 8; do echo "== mask_head16=$m"; SN_MASK16=$m timeout 300 python tools/mask_profile.py mask 2>&1 | grep ms; done; done > $out/mask16_ab.txt 2>&1; cat $out/mask16_ab.txt
SN_LIB=ab/exp_trace.so SN_MASK16=8 SN_TRACE_CHUNKS=27 timeout 300 python tools/mask_trace.py 2>&1 | grep -v amdgpu > $out/mask16_trace.txt; cat $out/mask16_trace.txt
