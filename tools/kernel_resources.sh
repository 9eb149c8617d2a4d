#!/bin/bash
# usage: tools/kernel_resources.sh [extra -D flags...]   -> VGPR/SGPR/spill/occupancy of the fused kernels of render.hip (no GPU needed)
cd "$(dirname "$0")/../sanerf-hq_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-function "$@" \
  -Rpass-analysis=kernel-resource-usage -c ${SN_SRC:-render.hip} -o /dev/null 2>&1 | python3 -c "
import re,sys
cur=None; rows=[]
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)',ln)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for k in ('VGPRs','AGPRs','SGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','SGPRs Spill','VGPRs Spill','LDS Size \[bytes/block\]'):
        m=re.search(k+r': (\d+)',ln)
        if m and cur is not None: cur[k.split(' [')[0].replace('\\\\','')]=int(m.group(1))
import subprocess
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    n=n.replace('void sn::','').replace('(sn::','(')[:86]
    print(f\"{n:88s} v={r.get('VGPRs')} a={r.get('AGPRs')} s={r.get('SGPRs')} scr={r.get('ScratchSize')} occ={r.get('Occupancy')} sspill={r.get('SGPRs Spill')} vspill={r.get('VGPRs Spill')}\")
"
