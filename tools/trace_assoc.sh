#!/bin/bash
# Kernel trace of the association bench: tools/trace_assoc.sh <tag>  ->  gpurun_out/<tag>_stats
TAG=${1:-assoc}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/${TAG}_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_stats -- python3 bench.py --config assoc --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_stats.log 2>&1
