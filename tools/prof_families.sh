cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in dpcknn_small_patch16_224 kmedoids_small_patch16_224 dyvit_small_patch16_224 ats_small_patch16_224 sit_small_patch16_224; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fam_$m -o stats -- python3 tools/run_model.py $m 0.7 256 20 > gpurun_out/fam_$m.log 2>&1
  grep images gpurun_out/fam_$m.log
done
