#!/bin/bash
# usage (GPU box, repo root): tools/ab_bench.sh <other-lib.so> [bench args...]
# Same-box A/B of two builds of the library on the bench line (box-to-box spread is +-5 %): B A B A, value + ms each.
other=$1; shift
for i in 1 2; do
  for lib in "$other" ""; do
    SN_LIB=$lib python bench.py --no-cpu-baseline --primary-only "$@" 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-HEAD}'.split('/')[-1], d['config']['schedule'], round(d['value']/1e6,2), 'Mrays/s', d['ms_per_step'], 'ms')"
  done
done
