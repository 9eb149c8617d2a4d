#!/bin/bash
# usage (GPU box, repo root): tools/xswap_ab.sh [out-file]   -- which lanes trade the two x-corners of a hashed cell in the last stage (SN_XSWAP_MODE, render.hip):
# bench line (800x800, [128], fp16 tables; and fp32 tables) of the product library and of ab/xswap1.so (neighbour lanes) / ab/xswap2.so (16-lane rows), built
# by tools/build_variant.sh, alternating, with a checksum of the image (the modes must agree bit for bit).
out=${1:-$GRAFT_REPO_ROOT/gpurun_out/r06/xswap_ab.txt}; mkdir -p $(dirname $out); : > $out
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in product xswap1 xswap2; do
    [ $lib = product ] && unset SN_LIB || export SN_LIB=$GRAFT_REPO_ROOT/ab/$lib.so
    for tab in f16 f32; do
      line=$(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only --tables $tab 2>/dev/null | tail -1)
      echo "$lib $tab $(echo $line | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"].get("avg_kernel_ms"))')" >> $out
    done
    python - >> $out <<'P'
import os, sys, hashlib, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from helpers import product_model, synthetic_params
from sanerf_hq_amd import raymarching as rm, synth
dev = torch.device("cuda:0")
for steps in ([128], [128, 64, 32]):
    model = product_model(synthetic_params(steps, seed=1), steps, False, dev)
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(400, 400), 400, 400, device=dev)
    for dt in (torch.float16, torch.float32):
        img = rm.render_rays(rm.RenderPlan(model, steps, dt), ro, rd, tile_w=400)["image"]
        print("  image sha", steps, str(dt)[-7:], hashlib.sha256(img.cpu().numpy().tobytes()).hexdigest()[:16])
P
  done
done
cat $out
