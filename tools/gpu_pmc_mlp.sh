#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc_mlp.sh   -> gpurun_out/pmc_head_mlp.txt : MfmaUtil / VALUBusy / wait counters of the head MLP kernel alone (tools/mlp_bench.py)
out=$GRAFT_REPO_ROOT/gpurun_out; root=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
pf=$out/pmc_head_mlp.txt; rm -f $pf
for j in 1 0; do
  export SN_WIDE_JIT=$j
  echo "==== SN_WIDE_JIT=$j ($([ $j = 1 ] && echo k_mlp_wide_j || echo k_mlp_wide)): SAM head MLP 160 000 rows (13 x 3+20 launches), mask MLP 131 072 rows" >> $pf
  while read -r c; do
    [ -z "$c" ] && continue
    rm -rf $out/_p; rocprofv3 --pmc $c --kernel-trace -d $out/_p -o pmc -- python $root/tools/mlp_bench.py > /dev/null 2>&1
    echo "== pass: $c" >> $pf
    python $root/tools/rocpd_summary.py pmc $out/_p/pmc_results.db | grep -E "k_mlp_wide" | grep -v k_pack >> $pf 2>&1
    rm -rf $out/_p
  done <<LIST
MfmaUtil VALUBusy
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
LIST
done
cat $pf
