#!/bin/bash
# round 5, GPU run 8: k_mask16 (background-unit tile pipeline) in the experiments build: parity, determinism, A/B against k_mlp_wide_j<3>, trace
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5; mkdir -p $out
export SN_LIB=sanerf-hq_amd/libsanerf_hip_exp.so
timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q -k "16_row" 2>&1 | tail -5
for i in 1 2 3; do for m in 0 8; do echo "== mask_head16=$m"; SN_MASK16=$m timeout 300 python tools/mask_profile.py mask 2>&1 | grep ms; done; done > $out/run8_ab.txt 2>&1; cat $out/run8_ab.txt
SN_LIB=ab/exp_trace.so SN_MASK16=8 SN_TRACE_CHUNKS=27 timeout 300 python tools/mask_trace.py 2>&1 | grep -v amdgpu > $out/run8_trace.txt; cat $out/run8_trace.txt
