#!/bin/bash
# Lab: compile csrc/tr_mlp_fused.hip alone and print registers / spills of every kernel in it (extra -D flags as arguments)
cd "$(dirname "$0")/../../tokenreduction_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-inline-asm -Wno-unused-const-variable -ffp-contract=fast \
  -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage "$@" -c ${MF_SRC:-tr_mlp_fused.hip} -o /tmp/mf_lab.o 2>&1 |
  grep -E "error|warning:|Function Name|VGPRs:|VGPRs Spill|ScratchSize" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//'
