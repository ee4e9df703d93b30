#!/bin/bash
# usage: traffic_passes.sh <tag> <bench args...> -- FETCH_SIZE and WRITE_SIZE passes (separate, --kernel-trace only) of bench.py,
# summarised by profiles/collect_traffic.py into gpurun_out/<tag>_traffic.json; plus the --stats kernel summary of the same command
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_f -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/${tag}_pmc_f.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc_w -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/${tag}_pmc_w.log 2>&1 &&
python3 profiles/collect_traffic.py gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w "$tag" > gpurun_out/${tag}_traffic.json &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/${tag}_stats.log 2>&1 &&
cp $(ls -t gpurun_out/${tag}_stats/*/*_kernel_stats.csv | head -1) gpurun_out/${tag}_kernel_stats.csv
