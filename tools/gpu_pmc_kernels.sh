#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc_kernels.sh [round-dir]   -> gpurun_out/<round-dir>/pmc_kernels_<workload>.txt
# Counter passes (rocprofv3 --pmc with --kernel-trace only, one pass per counter group) for the kernels the bench line's roofline block does
# not cover (round-4 verdict item 6): k_mask16 / k_mlp_wide (mask render), k_feat_stage (configs[2]), k_bin_scatter / k_bin_accum /
# k_linear_wgrad_mfma / k_mlp_wide<4> (configs[4] step), k_prop_stage with fp16 tables (reference schedule).
R=${1:-r06}; out=$GRAFT_REPO_ROOT/gpurun_out/$R; mkdir -p $out; root=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TA|TCC|SQ)_[A-Z0-9_]+(_sum)?\b" | sort -u > $out/counters_available.txt
run() {   # name, filter, command...
  local name=$1 filt=$2; shift 2
  local pf=$out/pmc_kernels_$name.txt; rm -f $pf
  while read -r c; do
    [ -z "$c" ] && continue
    rm -rf $out/_p; timeout 600 rocprofv3 --pmc $c --kernel-trace -d $out/_p -o pmc -- "$@" > /dev/null 2>&1
    echo "== pass: $c" >> $pf
    python $root/tools/rocpd_summary.py pmc $out/_p/pmc_results.db 2>&1 | grep -E "$filt" >> $pf
    rm -rf $out/_p
  done <<LIST
MfmaUtil VALUBusy
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCC_HIT_sum TCC_MISS_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
FETCH_SIZE
WRITE_SIZE
LIST
}
run mask_head "k_mask16|k_mlp_wide|k_final_stage|k_prop_stage" python $root/tools/mask_profile.py mask
run c3_sam_head "k_feat_stage|k_mlp_wide|k_final_stage" python $root/tools/c3_profile.py
run train_mask "k_bin_|k_linear_wgrad|k_mlp_wide|k_grid_forward|k_adam" python $root/tools/train_profile.py mask
run train_rgb "k_mlp_small|k_bin_|k_linear_wgrad|k_grid_forward|k_ray_composite|k_weights|k_proposal_loss|k_adam" python $root/tools/train_profile.py rgb
run ref_f16 "k_prop_stage|k_final_stage" python $root/bench.py --steps 2 --warmup 1 --schedule ref --tables f16 --no-cpu-baseline --primary-only
run flat128_f16 "k_final_stage" python $root/bench.py --steps 2 --warmup 1 --schedule flat128 --tables f16 --no-cpu-baseline --primary-only
python $root/tools/pmc_kernels_summary.py $out $out/latest_kernel_counters.json > $out/pmc_kernels_summary.txt 2>&1
cp $out/latest_kernel_counters.json $root/profiles/latest_kernel_counters.json      # bench.py (also.*.roofline.binding) reads it: same sources, fingerprint matches
ls -la $out
