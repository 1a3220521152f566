#!/bin/bash
# Lab: the library with the in-kernel clock probes compiled in (-DTR_DIAG_CLOCK: tr_gemm.hip, tr_mlp_fused.hip) -> tools/lab/libtr_clock.so
set -e
cs=tokenreduction_amd/csrc
fl="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-inline-asm -Wno-unused-const-variable -ffp-contract=fast -DTR_DIAG_CLOCK"
/opt/rocm/bin/hipcc $fl -c $cs/tr_gemm.hip -o /tmp/clock_tr_gemm.o
/opt/rocm/bin/hipcc $fl -fno-slp-vectorize -c $cs/tr_mlp_fused.hip -o /tmp/clock_tr_mlp_fused.o
others=$(ls $cs/*.o | grep -v "/tr_gemm.o" | grep -v "/tr_mlp_fused.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/clock_tr_gemm.o /tmp/clock_tr_mlp_fused.o $others -o tools/lab/libtr_clock.so
echo built tools/lab/libtr_clock.so
