#!/bin/bash
# usage (GPU box, repo root): tools/ab_wide.sh [pytest]   same-box A/B of the head-MLP kernels: SN_WIDE_JIT=0 (k_mlp_wide) vs 1 (k_mlp_wide_j)
if [ "$1" = "pytest" ]; then
  python -m pytest tests -m gpu -x -q -k "wide_mlp or mask_head or heads or config3 or mask_training or sam_distillation" 2>&1 | tail -15
fi
for i in 1 2; do
  for j in 0 1; do
    echo "== SN_WIDE_JIT=$j"
    SN_WIDE_JIT=$j python tools/mlp_bench.py 2>&1 | grep -v Warning
    SN_WIDE_JIT=$j python tools/mask_profile.py mask 2>&1 | grep "mask render"
  done
done
