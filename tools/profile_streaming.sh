#!/bin/bash
# Counter passes (HBM traffic, SQ) and a kernel trace of the streaming single launch at a given size:
#   tools/profile_streaming.sh r03 8388608 2p23
TAG=${1:-rXX}; M=${2:-8388608}; NAME=${3:-2p23}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
P=gpurun_out/${TAG}_stream_${NAME}
ARGS="--corr $M --steps 6 --warmup 2 --no-cpu-baseline --no-large-scan"
rm -rf ${P}_stats ${P}_fetch ${P}_write ${P}_sq
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d ${P}_stats -- python3 bench.py --corr $M --steps 30 --warmup 3 --no-cpu-baseline --no-large-scan > ${P}_stats.log 2>&1 || { echo "trace failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d ${P}_sq -- python3 bench.py $ARGS > ${P}_sq.log 2>&1 || { echo "sq failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d ${P}_fetch -- python3 bench.py $ARGS > ${P}_fetch.log 2>&1 || { echo "fetch failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d ${P}_write -- python3 bench.py $ARGS > ${P}_write.log 2>&1 || { echo "write failed"; exit 1; }
python3 profiles/summarize.py stats ${P}_stats > gpurun_out/${TAG}_stream_${NAME}_kernel_stats.csv
python3 profiles/summarize.py counters ${P}_sq > gpurun_out/${TAG}_pmc_sq_stream_${NAME}.json
python3 profiles/summarize.py pmc ${P}_fetch ${P}_write > gpurun_out/${TAG}_pmc_traffic_stream_${NAME}.json
rm -rf ${P}_stats ${P}_fetch ${P}_write ${P}_sq
echo "streaming profile $NAME done"
