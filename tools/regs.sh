#!/bin/bash
# usage: regs.sh <file.hip> [extra flags] -> per-kernel VGPR / spill / LDS table (NP=4 kernels only by default)
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-array-bounds -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/regs_tmp.o 2>&1 | grep -E "error|Function Name|VGPRs:|VGPRs Spill|LDS Size" | sed 's/.*remark: //' | paste - - - - | sed 's/\[-Rpass[^]]*\]//g' | sed 's/Function Name: //' | cut -c1-170
