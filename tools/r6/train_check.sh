#!/bin/bash
# round 6: new training operators -- tests, step timing, kernel trace of the RGB step
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r6; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
cd $root
python -m pytest tests/test_gpu_train_ops.py -x -q 2>&1 | tail -25 > $out/pytest_train_ops.txt
python -m pytest tests/test_gpu_render.py -x -q -k "training or distillation or c5 or train" 2>&1 | tail -12 > $out/pytest_train_fixtures.txt
python tools/train_profile.py rgb > $out/train_rgb_ms.txt 2>&1
python tools/train_profile.py rgb_noprop >> $out/train_rgb_ms.txt 2>&1
cd /tmp
for m in rgb rgb_noprop; do
rm -rf $out/_t; rocprofv3 --kernel-trace --stats -d $out/_t -o t -- python $root/tools/train_profile.py $m > $out/train_${m}_under_rocprof.log 2>&1
python $root/tools/rocpd_summary.py stats $out/_t/t_results.db > $out/kernel_stats_train_${m}.txt 2>&1
rm -rf $out/_t
done
