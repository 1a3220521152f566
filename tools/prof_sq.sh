#!/bin/bash
# SQ counter pass (own run, kernel trace only): MFMA busy cycles, wave cycles and the wait split per kernel
#   bash tools/prof_sq.sh r02_final      (on the GPU box, through gpurun)
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${TAG}_sq -o sq -- python3 bench.py --steps 3 --warmup 1 --no-extra > gpurun_out/${TAG}_sq.log 2>&1
ls gpurun_out/${TAG}_sq; tail -2 gpurun_out/${TAG}_sq.log
