#!/bin/bash
# Round-4 profile set (on the GPU box, through gpurun):  bash tools/prof_r04.sh <tag> [workloads...]
# For EVERY BASELINE config (and the reduction families whose kernels no config above exercises): rocprofv3 kernel stats and the two
# HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel trace only), condensed by tools/prof_summary.py into
#   gpurun_out/r04_profiles/<tag>_<workload>_{kernel_stats.csv,pmc_traffic.json}      (copy what is to be judged into profiles/)
TAG=${1:-r04}; shift
WL=${@:-"headline tome atsb_train dpcknnb_train kmedb384 sinkb384 dpcknn_small ats_small sit_small evit_small"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export TR_BENCH_IN_FLIGHT=1      # per-kernel numbers are taken one forward at a time (two in flight overlap launches: durations and counters of different kernels would mix)
mkdir -p gpurun_out/r04_profiles
for w in $WL; do
  case $w in
    headline)      CMD="bench.py --steps 20 --warmup 3 --no-extra"; PMC="bench.py --steps 3 --warmup 1 --no-extra";;
    tome)          CMD="tools/run_model.py tome_small_patch16_224 r16 256 10"; PMC="tools/run_model.py tome_small_patch16_224 r16 256 2";;
    atsb_train)    CMD="tools/train_step.py ats_base_patch16_224 128 5"; PMC="tools/train_step.py ats_base_patch16_224 128 1";;
    dpcknnb_train) CMD="tools/train_step.py dpcknn_base_patch16_224 128 5"; PMC="tools/train_step.py dpcknn_base_patch16_224 128 1";;
    kmedb384)      CMD="tools/run_model.py kmedoids_base_patch16_224 0.25 64 10 384"; PMC="tools/run_model.py kmedoids_base_patch16_224 0.25 64 2 384";;
    sinkb384)      CMD="tools/run_model.py sinkhorn_base_patch16_224 0.25 64 10 384"; PMC="tools/run_model.py sinkhorn_base_patch16_224 0.25 64 2 384";;
    *_small)       CMD="tools/run_model.py ${w}_patch16_224 0.7 256 10"; PMC="tools/run_model.py ${w}_patch16_224 0.7 256 2";;
  esac
  D=gpurun_out/${TAG}_${w}
  rocprofv3 --kernel-trace --stats --output-format csv -d ${D}_stats -o stats -- python3 $CMD > ${D}_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${D}_fetch -o fetch -- python3 $PMC > ${D}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d ${D}_write -o write -- python3 $PMC > ${D}_write.log 2>&1
  S=$(find ${D}_stats -name "*kernel_stats.csv" | head -1); F=$(find ${D}_fetch -name "*counter_collection.csv" | head -1); W=$(find ${D}_write -name "*counter_collection.csv" | head -1)
  echo "== $w: $(tail -1 ${D}_stats.log | cut -c1-200)"
  python3 tools/prof_summary.py ${TAG}_${w} $S $F $W gpurun_out/r04_profiles | head -40
  rm -rf ${D}_fetch ${D}_write ${D}_stats      # raw traces are large; the condensed files are what travels back
done
