#!/bin/bash
# Round profile: bench line + rocprofv3 kernel stats + the two HBM-traffic PMC passes (separate runs, kernel trace only).
#   bash tools/prof_round.sh r02_final       (on the GPU box, through gpurun)
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -o stats -- python3 bench.py --steps 20 --warmup 3 --no-extra > gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_fetch -o fetch -- python3 bench.py --steps 3 --warmup 1 --no-extra > gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_write -o write -- python3 bench.py --steps 3 --warmup 1 --no-extra > gpurun_out/${TAG}_write.log 2>&1
ls -R gpurun_out/${TAG}_* | head -30
tail -c 3000 gpurun_out/${TAG}_bench.json
