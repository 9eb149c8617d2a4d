#!/bin/bash
# round 5, GPU run 14: split-fp16 training forward as the default + forward_cat in training: whole GPU suite, step times, kernel stats
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -q -x > $out/pytest_gpu_f16x3fwd.txt 2>&1; tail -4 $out/pytest_gpu_f16x3fwd.txt
timeout 900 python tools/bench_configs.py > $out/bench_configs_f16x3fwd.json 2> $out/bench_configs_f16x3fwd.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r5/bench_configs_f16x3fwd.json"))
for k, v in d.items():
    if "train" in k.lower() or "C5" in k:
        print(k, json.dumps(v)[:600])
PY
python tools/r5/fwd_modes_err.py 2>&1 | grep "rel-L2" > $out/fwd_modes_err.txt; cat $out/fwd_modes_err.txt
python tools/mlp_f32_bench.py 2>&1 | grep -v amdgpu | tail -4 > $out/mlp_fwd_bench.txt; cat $out/mlp_fwd_bench.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/_t; rocprofv3 --kernel-trace --stats -d /tmp/_t -o t -- python $GRAFT_REPO_ROOT/tools/train_profile.py mask > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats /tmp/_t/t_results.db > $GRAFT_REPO_ROOT/$out/kernel_stats_train_mask_f16x3fwd.txt 2>&1; head -22 $GRAFT_REPO_ROOT/$out/kernel_stats_train_mask_f16x3fwd.txt | cut -c1-150
