#!/bin/bash
# Round-5 busy-unit profile set (on the GPU box, through gpurun):  bash tools/prof_r05_sq.sh <tag> [workloads...]
# VERDICT r04 item 5 / north_star "MFMA utilisation against gfx950 peak": for every BASELINE config's workload ONE rocprofv3 --pmc pass
# (kernel trace only, the program itself behind `--`) with the SQ counters, condensed per kernel by tools/prof_r05_sq_summary.py into
#   gpurun_out/r05_profiles/<tag>_pmc_sq_<workload>.json      (copy what is to be judged into profiles/)
TAG=${1:-r05}; shift
WL=${@:-"headline atsb_train kmedb384 sinkb384 dpcknn_small tome"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export TR_BENCH_IN_FLIGHT=1      # per-kernel numbers are taken one forward at a time (two in flight overlap launches: durations and counters of different kernels would mix)
mkdir -p gpurun_out/r05_profiles
for w in $WL; do
  case $w in
    headline)      PMC="bench.py --steps 3 --warmup 1 --no-extra";;
    tome)          PMC="tools/run_model.py tome_small_patch16_224 r16 256 2";;
    atsb_train)    PMC="tools/train_step.py ats_base_patch16_224 128 1";;
    dpcknnb_train) PMC="tools/train_step.py dpcknn_base_patch16_224 128 1";;
    kmedb384)      PMC="tools/run_model.py kmedoids_base_patch16_224 0.25 64 2 384";;
    sinkb384)      PMC="tools/run_model.py sinkhorn_base_patch16_224 0.25 64 2 384";;
    *_small)       PMC="tools/run_model.py ${w}_patch16_224 0.7 256 2";;
  esac
  D=gpurun_out/${TAG}_${w}_sq
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
            --kernel-trace --output-format csv -d ${D}a -o sq -- python3 $PMC > ${D}a.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
            --kernel-trace --output-format csv -d ${D}b -o sq -- python3 $PMC > ${D}b.log 2>&1
  A=$(find ${D}a -name "*counter_collection.csv" | head -1); B=$(find ${D}b -name "*counter_collection.csv" | head -1)
  echo "== $w: $(tail -1 ${D}a.log | cut -c1-160)"
  python3 tools/prof_r05_sq_summary.py ${TAG} $w gpurun_out/r05_profiles "$A" "$B" | head -30
  rm -rf ${D}a ${D}b
done
