#!/bin/bash
# Round-3 profile set (on the GPU box, through gpurun):  bash tools/prof_r03.sh <tag>
#   headline: bench line, rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE passes, SQ pass (each its own run: kernel trace only with --pmc)
#   training step (topk_small): kernel stats + SQ pass
#   the other BASELINE configs: kernel stats of configs[2] ToMe r16 eval, configs[3] ATS-B train step, configs[4] K-Medoids-B 384^2 eval
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -o stats -- python3 bench.py --steps 20 --warmup 3 --no-extra > gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_fetch -o fetch -- python3 bench.py --steps 3 --warmup 1 --no-extra > gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_write -o write -- python3 bench.py --steps 3 --warmup 1 --no-extra > gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${TAG}_sq -o sq -- python3 bench.py --steps 3 --warmup 1 --no-extra > gpurun_out/${TAG}_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_train_stats -o stats -- python3 tools/train_step.py topk_small_patch16_224 256 5 > gpurun_out/${TAG}_train_stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${TAG}_train_sq -o sq -- python3 tools/train_step.py topk_small_patch16_224 256 2 > gpurun_out/${TAG}_train_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_tome_stats -o stats -- python3 tools/run_model.py tome_small_patch16_224 r16 256 10 > gpurun_out/${TAG}_tome_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_atsb_train_stats -o stats -- python3 tools/train_step.py ats_base_patch16_224 128 5 > gpurun_out/${TAG}_atsb_train_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_kmedb384_stats -o stats -- python3 tools/run_model.py kmedoids_base_patch16_224 0.25 64 10 384 > gpurun_out/${TAG}_kmedb384_stats.log 2>&1
for f in stats train_stats tome_stats atsb_train_stats kmedb384_stats; do tail -1 gpurun_out/${TAG}_${f}.log; done
find gpurun_out/${TAG}_* -name "*.csv" | head -40
tail -c 600 gpurun_out/${TAG}_bench.json
