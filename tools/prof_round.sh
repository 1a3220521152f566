cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r01_final_bench.json 2> gpurun_out/r01_final_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01_final_stats -o stats -- python3 bench.py --steps 20 --warmup 3 --no-extra > gpurun_out/r01_final_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r01_final_fetch -o fetch -- python3 bench.py --steps 3 --warmup 1 --no-extra > gpurun_out/r01_final_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r01_final_write -o write -- python3 bench.py --steps 3 --warmup 1 --no-extra > gpurun_out/r01_final_write.log 2>&1
ls -R gpurun_out/r01_final_* | head -30
tail -c 1500 gpurun_out/r01_final_bench.json
