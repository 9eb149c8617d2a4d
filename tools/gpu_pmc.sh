#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc.sh <schedule> <outdir> ; runs several rocprofv3 --pmc passes (counters only + kernel-trace)
sch=${1:-ref}; out=$GRAFT_REPO_ROOT/gpurun_out/${2:-pmc}; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
while read -r c; do
  [ -z "$c" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -d $out/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --schedule $sch --no-cpu-baseline --primary-only > $out/p$i.log 2>&1
  echo "== pass $i: $c" >> $out/summary.txt
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py pmc $out/p$i/pmc_results.db >> $out/summary.txt 2>&1
  rm -rf $out/p$i
done <<LIST
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TAGRAM0_REQ_sum
TCC_HIT_sum TCC_MISS_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
FETCH_SIZE
MfmaUtil VALUBusy
OccupancyPercent MeanOccupancyPerCU
LIST
cat $out/summary.txt
