#!/bin/bash
# usage (GPU box, repo root): tools/ab_env.sh "<bench args>" "VAR=a" "VAR=b" ...   same-box comparison of environment switches of the library
args=$1; shift
for i in 1 2; do
  for kv in "$@"; do
    env $kv python bench.py --no-cpu-baseline --primary-only $args 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; o=r['other_kernels_ms']
print('%-24s %-8s %7.2f Mrays/s %7.3f ms | final %.3f prop0 %s prop1 %s | clk %s MHz' % ('$kv', d['config']['schedule'], d['value']/1e6, d['ms_per_step'], r['avg_kernel_ms'], o['prop0'], o['prop1'], r['shader_clock_mhz']))"
  done
done
