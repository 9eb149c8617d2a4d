#!/bin/bash
# round 5, GPU run 12: binned grid backward with reference entries (k_bin_refs / k_bin_pull), per-block slot matrix, new bin geometry:
# whole GPU suite + the training-step benches + kernel stats of the mask-field step
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -q -x > $out/pytest_gpu_binpull.txt 2>&1; tail -4 $out/pytest_gpu_binpull.txt
timeout 900 python tools/bench_configs.py > $out/bench_configs_binpull.json 2> $out/bench_configs_binpull.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r5/bench_configs_binpull.json"))
for k, v in d.items():
    if "train" in k.lower() or "C5" in k:
        print(k, json.dumps(v)[:600])
PY
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/_t; rocprofv3 --kernel-trace --stats -d /tmp/_t -o t -- python $GRAFT_REPO_ROOT/tools/train_profile.py mask > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats /tmp/_t/t_results.db > $GRAFT_REPO_ROOT/$out/kernel_stats_train_mask_binpull.txt 2>&1; head -30 $GRAFT_REPO_ROOT/$out/kernel_stats_train_mask_binpull.txt | cut -c1-150
