#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc_prop_lds.sh [round-dir]  -> gpurun_out/<round-dir>/pmc_prop_stage_lds.txt
# What do the waves of k_prop_stage wait for?  LDS-side counters of the reference schedule (800x800, fp16 tables): the stage reads its 176 MLP weights
# per sample as 44 broadcast ds_read_b128 (1 KiB each through a 128 B/clk pipe).
R=${1:-r06}; out=$GRAFT_REPO_ROOT/gpurun_out/$R; mkdir -p $out; root=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
pf=$out/pmc_prop_stage_lds.txt; rm -f $pf
while read -r c; do
  [ -z "$c" ] && continue
  rm -rf $out/_p; timeout 600 rocprofv3 --pmc $c --kernel-trace -d $out/_p -o pmc -- python $root/bench.py --steps 2 --warmup 1 --schedule ref --tables f16 --no-cpu-baseline --primary-only > /dev/null 2>&1
  echo "== pass: $c" >> $pf
  python $root/tools/rocpd_summary.py pmc $out/_p/pmc_results.db 2>&1 | grep -E "k_prop_stage|k_final_stage|kernel" >> $pf
  rm -rf $out/_p
done <<LIST
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU
LIST
cat $pf
