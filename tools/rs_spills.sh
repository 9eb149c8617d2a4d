#!/bin/bash
# usage: tools/rs_spills.sh [-D...]   -> where the scratch spills of k_final_stage_rs<float> sit (fast path / slow path of the producer, consumer)
cd "$(dirname "$0")/../sanerf-hq_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -Wno-unused-function "$@" -S --cuda-device-only render.hip -o /tmp/render_rs.s 2>/dev/null
for ty in f 6__half; do
awk "/^_ZN2sn16k_final_stage_rsI${ty}Li5EEEvNS_9FinalArgsE:/,/s_endpgm/" /tmp/render_rs.s > /tmp/rs_$ty.s
python3 - /tmp/rs_$ty.s $ty <<'PY'
import re, sys
lines = open(sys.argv[1]).read().split('\n')
# main loops: a "Loop Header: Depth=1" that contains s_sleep children -> producer sample loop / consumer loop
hdr = [i for i, l in enumerate(lines) if 'Loop Header: Depth=1' in l and not 'Inner' in l]
sc = [i for i, l in enumerate(lines) if 'scratch_' in l]
print(sys.argv[2], 'lines', len(lines), 'scratch ops', len(sc), 'main-loop headers at', hdr, ' v_readlane', sum('v_readlane' in l for l in lines), 'v_writelane', sum('v_writelane' in l for l in lines))
for i in sc[:60]:
    print('   ', i, lines[i].strip()[:90])
PY
done
