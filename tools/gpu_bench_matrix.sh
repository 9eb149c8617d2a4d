#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_bench_matrix.sh "<modes>" "<schedules>" "<tables>"
mkdir -p gpurun_out
for m in $1; do for sch in $2; do for tb in ${3:-f32}; do
  timeout 300 python bench.py --tuning mlp_mode=$m --steps 10 --warmup 3 --schedule $sch --tables $tb --no-cpu-baseline > gpurun_out/bench_${m}_${sch}_${tb}.log 2>&1
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_${m}_${sch}_${tb}.log").read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"$m $sch $tb: {d['value']/1e6:8.2f} Mrays/s  {d['ms_per_step']:7.3f} ms/step | final {r['avg_kernel_ms']} ms, others {r['other_kernels_ms']} | hbm-frac final {r['frac']} whole {r['whole_path']['frac']}")
except Exception as e:
    print("$m $sch $tb: FAILED", e); print(open("gpurun_out/bench_${m}_${sch}_${tb}.log").read()[-800:])
PY
done; done; done
