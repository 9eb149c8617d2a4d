#!/bin/bash
# usage: tools/kernel_spills.sh <mangled-name-substring> [-D...]   -> scratch spill/reload lines of one kernel of render.hip with the
# loop headers around them (is a spill inside the hot loop or in a prologue / slow path?)
pat=$1; shift
cd "$(dirname "$0")/../sanerf-hq_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -Wno-unused-function "$@" -S --cuda-device-only render.hip -o /tmp/render_ks.s 2>/dev/null
python3 - "$pat" <<'PY'
import re, sys
src = open('/tmp/render_ks.s').read().split('\n')
pat = sys.argv[1]
start = [i for i, l in enumerate(src) if re.match(r'^_Z\w+:', l) and pat in l]
for s0 in start:
    e = next(i for i in range(s0, len(src)) if 's_endpgm' in src[i])
    body = src[s0:e]
    print(src[s0][:120], 'lines', len(body))
    for i, l in enumerate(body):
        if 'Loop Header' in l or 'scratch_' in l:
            print('  ', i, l.strip()[:110])
PY
