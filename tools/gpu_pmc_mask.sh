#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc_mask.sh   -> gpurun_out/pmc_mask_head.txt : counters of k_mlp_wide in the 400x400 mask render
out=$GRAFT_REPO_ROOT/gpurun_out; root=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
pf=$out/pmc_mask_head.txt; rm -f $pf
while read -r c; do
  [ -z "$c" ] && continue
  rm -rf $out/_p; rocprofv3 --pmc $c --kernel-trace -d $out/_p -o pmc -- python $root/tools/mask_profile.py mask > /dev/null 2>&1
  echo "== pass: $c" >> $pf
  python $root/tools/rocpd_summary.py pmc $out/_p/pmc_results.db | grep -E "k_mlp_wide|k_feat|k_final" >> $pf 2>&1
  rm -rf $out/_p
done <<LIST
MfmaUtil VALUBusy
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum
TCC_HIT_sum TCC_MISS_sum
LIST
cat $pf
