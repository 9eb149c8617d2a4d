#!/bin/bash
# usage (GPU box): tools/bin_stats.sh <out.txt> [lib.so ...]  -- per-kernel times of the binned grid backward (tools/bin_trace.py: mask grid, C = 8,
# 4096 rays x 32 samples) under rocprofv3 --kernel-trace --stats, one block per library ("" = the tree's build)
cd /tmp; export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$1; shift; mkdir -p $(dirname $out); : > $out
for lib in "${@:-}"; do
  echo "== lib=${lib:-HEAD}" >> $out
  rm -rf /tmp/_bt; SN_LIB=${lib:+$root/$lib} rocprofv3 --kernel-trace --stats -d /tmp/_bt -o t -- python $root/tools/bin_trace.py ${BIN_C:-8} > /tmp/_bt.log 2>&1
  grep "backward" /tmp/_bt.log >> $out
  python $root/tools/rocpd_summary.py stats /tmp/_bt/t_results.db 2>&1 | grep -i "k_bin\|kernel  \|grid\|fill" | cut -c1-150 >> $out
done
cat $out
