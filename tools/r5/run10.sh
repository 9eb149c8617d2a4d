#!/bin/bash
# round 5, GPU run 10: k_mask16 as the product mask head: whole GPU suite, A/B in the experiments build, trace
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -q -x > $out/pytest_gpu_product.txt 2>&1; tail -4 $out/pytest_gpu_product.txt
export SN_LIB=sanerf-hq_amd/libsanerf_hip_exp.so
timeout 1200 python -m pytest tests -m gpu -q -k "experiments or just_in_time or mask16 or role or lds_level or wide_ab or narrow" > $out/pytest_gpu_experiments_build.txt 2>&1; tail -3 $out/pytest_gpu_experiments_build.txt
for i in 1 2 3; do for m in 0 8; do echo "== mask_head16=$m"; SN_MASK16=$m timeout 300 python tools/mask_profile.py mask 2>&1 | grep ms; done; done > $out/mask16_ab.txt 2>&1; cat $out/mask16_ab.txt
