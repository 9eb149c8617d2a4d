#!/bin/bash
# round 5, GPU run 2: k_mlp16<3> (fused mask head on 16-row tiles, two waves per SIMD): layout probe, parity, same-box A/B, cycle trace
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5; mkdir -p $out
tools/ubench/mfma16_layout_ub > $out/run2_layout.txt 2>&1; cat $out/run2_layout.txt
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_ops.py tests/test_gpu_render.py tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q \
  -k "mask or head" > $out/run2_pytest.txt 2>&1
tail -25 $out/run2_pytest.txt
for i in 1 2; do for lib in "" ab/pairs0.so; do echo "== lib=${lib:-HEAD}"; SN_LIB=$lib timeout 300 python tools/mask_profile.py mask; done; done > $out/run2_ab.txt 2>&1
cat $out/run2_ab.txt
for lib in ab/wtrace.so; do echo "== $lib"; SN_LIB=$lib timeout 300 python tools/mask_trace.py; done > $out/run2_trace.txt 2>&1
cat $out/run2_trace.txt
