#!/bin/bash
# round 5, GPU run 4: k_mlp16 with the pipelined tile hand-over: parity, A/B (product, no-BB-split, round-4 kernel), trace
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_ops.py tests/test_gpu_render.py tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q \
  -k "mask or head or packed" > $out/run4_pytest.txt 2>&1
tail -8 $out/run4_pytest.txt
for i in 1 2; do for lib in "" ab/nobb.so ab/old3.so; do echo "== lib=${lib:-HEAD}"; SN_LIB=$lib timeout 300 python tools/mask_profile.py mask; done; done > $out/run4_ab.txt 2>&1
grep -v amdgpu.ids $out/run4_ab.txt
for lib in ab/wtrace.so; do echo "== $lib"; SN_LIB=$lib timeout 300 python tools/mask_trace.py; done > $out/run4_trace.txt 2>&1
grep -v amdgpu.ids $out/run4_trace.txt
