cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_backward.py tests/test_hip_train.py -x -q -m gpu 2>&1 | tail -4
python tools/train_step.py topk_small_patch16_224 256 5 2>&1 | tail -1
python tools/train_step.py deit_small_patch16_224_local 256 5 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_train_stats -o stats -- python3 tools/train_step.py topk_small_patch16_224 256 5 > gpurun_out/r02_train_stats.log 2>&1
