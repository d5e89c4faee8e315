#!/bin/bash
# usage: tools/resource_usage.sh file.hip [extra hipcc flags]  -> per-kernel VGPR / SGPR / LDS / occupancy / scratch summary
f=$1; shift
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 \
 | grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | c++filt \
 | awk '/Function Name:/ {if (n!="") print n" ::"l; sub(/.*Function Name: /,""); n=$0; l=""; next} {gsub(/^ +/,""); l=l" | "$0} END{print n" ::"l}' \
 | sed -E 's/TotalSGPRs/SGPR/; s/Occupancy \[waves\/SIMD\]/Occ/; s/LDS Size \[bytes\/block\]/LDS/; s/ScratchSize \[bytes\/lane\]/Scratch/; s/ \| Dynamic Stack: False//; s/\(anonymous namespace\):://g; s/ \| AGPRs: 0//; s/ \| wavefrontsize64//'
