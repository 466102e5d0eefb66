#!/bin/bash
# Collects the round's evidence on a GPU box, into gpurun_out/<tag>_*:
#   tools/profile_round.sh r01d
# a plain bench line (with the CPU baseline), a kernel trace with statistics of the same command, and
# the two HBM-traffic counter passes (each in a run of its own, without any other tracing).
set -e
TAG=${1:-rXX}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write
timeout -k 10 300 python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-large-scan"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_stats -- python3 bench.py $ARGS > gpurun_out/${TAG}_stats.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_fetch -- python3 bench.py $ARGS > gpurun_out/${TAG}_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_write -- python3 bench.py $ARGS > gpurun_out/${TAG}_write.log 2>&1
echo done
