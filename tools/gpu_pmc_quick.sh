#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc_quick.sh <schedule> <outdir> [tables]; a few PMC passes
sch=${1:-flat128}; out=$GRAFT_REPO_ROOT/gpurun_out/${2:-pmcq}; tb=${3:-f32}; mkdir -p $out; rm -f $out/summary.txt
cd /tmp; export TMPDIR=/tmp
i=0
while read -r c; do
  [ -z "$c" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -d $out/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --schedule $sch --tables $tb --no-cpu-baseline --primary-only > $out/p$i.log 2>&1
  echo "== pass $i: $c" >> $out/summary.txt
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py pmc $out/p$i/pmc_results.db >> $out/summary.txt 2>&1
  rm -rf $out/p$i $out/p$i.log
done <<LIST
MfmaUtil VALUBusy
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum
TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum
TCC_HIT_sum TCC_MISS_sum
FETCH_SIZE
WRITE_SIZE
LIST
grep -v k_pack $out/summary.txt
