#!/bin/bash
# Lab: tr_lnlin_bf16 with parts ablated (timing only: outputs of the ablated builds are wrong by design)
for v in "" NO_LN NO_STORE NO_DMA NO_MFMA; do
  lib=tokenreduction_amd/csrc/libtokenreduction_hip.so
  [ -n "$v" ] && lib=tools/lab/libtr_ll_$v.so
  echo "== ${v:-product}"
  TOKENREDUCTION_HIP_LIB=$PWD/$lib python tools/lab/lnlin_lab.py 2>&1 | grep "M="
done
