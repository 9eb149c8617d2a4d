#!/bin/bash
# round 5, GPU run 7: the whole GPU suite on the product library, then the experiments-build tests
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -q -x > $out/pytest_gpu_product.txt 2>&1; tail -6 $out/pytest_gpu_product.txt
SN_LIB=sanerf-hq_amd/libsanerf_hip_exp.so timeout 1200 python -m pytest tests -m gpu -q -k "experiments or just_in_time or 16_row or role or lds_level or wide_ab or narrow" > $out/pytest_gpu_experiments_build.txt 2>&1; tail -4 $out/pytest_gpu_experiments_build.txt
