#!/bin/bash
# usage (GPU box): tools/bin_pmc.sh <out.txt> [lib.so]  -- counter passes (rocprofv3 --pmc + --kernel-trace only, one pass per group) of the binned
# grid backward's kernels at the mask-field step's size (tools/bin_trace.py)
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$1; lib=$2; mkdir -p $(dirname $out); : > $out; cd /tmp; export TMPDIR=/tmp
while read -r c; do
  [ -z "$c" ] && continue
  rm -rf /tmp/_bp; SN_LIB=${lib:+$root/$lib} timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/_bp -o pmc -- python $root/tools/bin_trace.py ${BIN_C:-8} > /dev/null 2>&1
  echo "== pass: $c" >> $out
  python $root/tools/rocpd_summary.py pmc /tmp/_bp/pmc_results.db 2>&1 | grep -E "k_bin_" >> $out
done <<LIST
VALUBusy GRBM_GUI_ACTIVE SQ_WAVES
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS_ATOMIC SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
FETCH_SIZE
WRITE_SIZE
LIST
cat $out
