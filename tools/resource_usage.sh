#!/bin/bash
# usage: tools_resource_usage.sh file.hip  -> per-kernel VGPR/AGPR/SGPR/LDS/occupancy/spill summary
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | c++filt \
 | awk '/Function Name:/ {if (n!="") print n" ::"l; sub(/.*Function Name: /,""); n=$0; l=""; next} {gsub(/^ +/,""); l=l" | "$0} END{print n" ::"l}' \
 | sed -E 's/TotalSGPRs/SGPR/; s/Occupancy \[waves\/SIMD\]/Occ/; s/LDS Size \[bytes\/block\]/LDS/; s/ScratchSize \[bytes\/lane\]/Scratch/; s/ \| Dynamic Stack: False//; s/\(anonymous namespace\):://'
