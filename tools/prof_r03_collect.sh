#!/bin/bash
# Copy the summaries of a tools/prof_r03.sh run from gpurun_out/ (scratch) into profiles/ (committed):  bash tools/prof_r03_collect.sh <tag>
TAG=${1:-r03}
python tools/prof_summary.py ${TAG} gpurun_out/${TAG}_stats/stats_kernel_stats.csv gpurun_out/${TAG}_fetch/fetch_counter_collection.csv gpurun_out/${TAG}_write/write_counter_collection.csv | tail -4
python tools/prof_sq_summary.py ${TAG} gpurun_out/${TAG}_sq/sq_counter_collection.csv | head -8
python tools/prof_sq_summary.py ${TAG}_train gpurun_out/${TAG}_train_sq/sq_counter_collection.csv | head -12
cp gpurun_out/${TAG}_train_stats/stats_kernel_stats.csv profiles/${TAG}_train_topk_small_kernel_stats.csv
cp gpurun_out/${TAG}_tome_stats/stats_kernel_stats.csv profiles/${TAG}_tome_small_r16_kernel_stats.csv
cp gpurun_out/${TAG}_atsb_train_stats/stats_kernel_stats.csv profiles/${TAG}_ats_base_train_kernel_stats.csv
cp gpurun_out/${TAG}_kmedb384_stats/stats_kernel_stats.csv profiles/${TAG}_kmedoids_base_384_kernel_stats.csv
tail -1 gpurun_out/${TAG}_bench.json > profiles/${TAG}_bench.json
ls -la profiles/${TAG}_*
