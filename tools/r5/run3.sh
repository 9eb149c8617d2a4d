#!/bin/bash
# round 5, GPU run 3: k_mlp16<3,4> (two 256-thread workgroups per CU) parity + A/B + trace; packed outputs; bench.py N>1 branch on one GPU
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_ops.py tests/test_gpu_render.py tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q \
  -k "mask or head or packed or bench_multi" > $out/run3_pytest.txt 2>&1
tail -25 $out/run3_pytest.txt
for i in 1 2; do for lib in "" ab/old3.so; do echo "== lib=${lib:-HEAD}"; SN_LIB=$lib timeout 300 python tools/mask_profile.py mask; done; done > $out/run3_ab.txt 2>&1
cat $out/run3_ab.txt
for lib in ab/wtrace.so; do echo "== $lib"; SN_LIB=$lib timeout 300 python tools/mask_trace.py; done > $out/run3_trace.txt 2>&1
cat $out/run3_trace.txt
timeout 600 python tools/tile_shape_ab.py > $out/tile_shape_ab.txt 2>&1; cat $out/tile_shape_ab.txt
