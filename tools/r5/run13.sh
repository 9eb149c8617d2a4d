#!/bin/bash
# round 5, GPU run 13: native fp32 forward of the training MLP (k_mlp_f32_train) + windowed k_bin_refs: parity, gradient fixtures, step time
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q -x -k "wide_mlp or binned or grid or train or c5 or gradient or fixture or mask_nll or graph" > $out/run13_pytest.txt 2>&1; tail -5 $out/run13_pytest.txt
python tools/grid_bwd_bench.py 2>&1 | grep -v amdgpu.ids | sed -e "s/(incl. the zero-fill of the gradient table)//g" -e "s/atomic.*//" | cut -c1-170
timeout 900 python tools/bench_configs.py > $out/bench_configs_f32fwd.json 2> $out/bench_configs_f32fwd.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r5/bench_configs_f32fwd.json"))
for k, v in d.items():
    if "train" in k.lower() or "C5" in k:
        print(k, json.dumps(v)[:600])
PY
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/_t; rocprofv3 --kernel-trace --stats -d /tmp/_t -o t -- python $GRAFT_REPO_ROOT/tools/train_profile.py mask > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats /tmp/_t/t_results.db > $GRAFT_REPO_ROOT/$out/kernel_stats_train_mask_f32fwd.txt 2>&1; head -24 $GRAFT_REPO_ROOT/$out/kernel_stats_train_mask_f32fwd.txt | cut -c1-150
