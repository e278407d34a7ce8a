#!/bin/bash
# device-only compile of one stage's kernels with the resource-usage remarks: VGPRs / occupancy / spills of the kernels named by $1 (a regex);
# $2 = the translation unit (default kamd_match; kamd_ec, kamd_em, kamd_io, kamd_ctx)
TU=${2:-kamd_match}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function --cuda-device-only -Rpass-analysis=kernel-resource-usage -c /root/repo/kallisto_amd/csrc/$TU.hip -o /tmp/kk_dev.o 2> /tmp/kk_usage.txt
grep -E "error|warning: " /tmp/kk_usage.txt | head
grep -A9 -E "Function Name: .*($1)" /tmp/kk_usage.txt | grep -E "Function Name|VGPRs:|Occupancy|Spill|ScratchSize" | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' | paste - - - - - - | cut -c1-260
