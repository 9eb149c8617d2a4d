#!/bin/bash
# same-box A/B of library builds on small linear-order batches (whole fused render and C5 step): tools/final_sp_ab.sh ab/x.so ab/y.so
for lib in "" "$@"; do
  echo "== ${lib:-HEAD}"
  SN_LIB=$lib python tools/prop_sp_lanes_ab.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print({k: v['lanes_auto']['render_ms'] for k,v in d.items()})"
  SN_LIB=$lib python tools/train_bench.py mask 2>/dev/null | tail -1
done
