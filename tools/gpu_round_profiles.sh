#!/bin/bash
# usage (GPU box, repo root): tools/gpu_round_profiles.sh [round-dir]  -- every profile artefact of a round from the code as it is:
# the counter passes of all hot kernels first (their JSON twin feeds bench.py's also.*.roofline.binding), then kernel-trace summaries,
# the bench kernels' PMC passes (latest_traffic.json), micro-benchmarks, robustness runs and the un-profiled bench lines.
R=${1:-r06}
bash $GRAFT_REPO_ROOT/tools/gpu_pmc_kernels.sh $R > $GRAFT_REPO_ROOT/gpurun_out/pmc_kernels_$R.log 2>&1
bash $GRAFT_REPO_ROOT/tools/gpu_profile_all.sh $R > $GRAFT_REPO_ROOT/gpurun_out/profile_all_$R.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/profile_all_$R.log
