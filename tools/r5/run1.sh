#!/bin/bash
# round 5, GPU run 1: parity of the pair-interleaved wide-MLP chunks + advisor fixes, same-box A/B against the one-tile-at-a-time form
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_ops.py tests/test_gpu_render.py tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -x -q \
  -k "mask or head or wide or mlp or binned or wgrad" > $out/run1_pytest.txt 2>&1
tail -5 $out/run1_pytest.txt
for i in 1 2; do for lib in "" ab/pairs0.so; do echo "== lib=${lib:-HEAD}"; SN_LIB=$lib timeout 300 python tools/mask_profile.py mask; SN_LIB=$lib timeout 300 python tools/mlp_bench.py; done; done > $out/run1_ab.txt 2>&1
cat $out/run1_ab.txt
for lib in ab/wtrace.so ab/wtrace0.so; do echo "== $lib"; SN_LIB=$lib timeout 300 python tools/mask_trace.py; done > $out/run1_trace.txt 2>&1
cat $out/run1_trace.txt
