#!/bin/bash
# kernel resource usage of one csrc/*.hip file (VGPR/AGPR/spills/occupancy); keeps the .s in /tmp
f=${1:-render.hip}
cd "/root/repo/sanerf-hq_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage -save-temps=obj -c $f -o /tmp/${f%.hip}_ra.o 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs|Spill|Occupancy|ScratchSize|LDS Size" | sed 's/.*remark: *//; s/ \[-Rpass.*//; s/[a-z_]*.hip:[0-9]*:0: *//' | paste - - - - - - - - | sed 's/Function Name: _ZN2sn[0-9]*//; s/ScratchSize \[bytes\/lane\]/Scratch/; s/Occupancy \[waves\/SIMD\]/Occ/; s/LDS Size \[bytes\/block\]/LDS/' | cut -c1-220
