#!/bin/bash
# Lab: the fused Mlp kernel's ablation builds (tools/lab/build_variant.sh mf_<name> ...), each ALSO with -DTR_DIAG_STAMPS; logs to gpurun_out/
for n in "$@"; do
  echo "== $n"
  TOKENREDUCTION_HIP_LIB=tools/lab/libtr_mf_$n.so timeout 120 python tools/mlp_lab.py --stamps 32768 50432 2>&1 | grep "M=\|stamps\|clock"
done
