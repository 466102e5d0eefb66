#!/bin/bash
# Collects the round's evidence on a GPU box, into gpurun_out/<tag>_*:
#   tools/profile_round.sh r02a [loop|batched|assoc ...]
# per configuration: a plain bench line (with the CPU baseline), a kernel trace with statistics of the
# same command, and the counter passes (HBM traffic, SQ instruction mix), each in a run of its own
# without any other tracing.  profiles/summarize.py turns the directories into the files kept under profiles/.
TAG=${1:-rXX}
shift
CONFIGS=${@:-loop}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for CFG in $CONFIGS; do
  P=gpurun_out/${TAG}_${CFG}
  rm -rf ${P}_stats ${P}_fetch ${P}_write ${P}_sq
  timeout -k 10 400 python3 bench.py --config $CFG > ${P}_bench.json 2> ${P}_bench.err || { echo "bench $CFG failed"; exit 1; }
  ARGS="--config $CFG --steps 10 --warmup 2 --no-cpu-baseline --no-large-scan"
  [ "$CFG" = assoc ] && ARGS="--config $CFG --steps 4 --warmup 1 --no-cpu-baseline"
  # the trace runs the default number of steps (a run of a dozen loops ends before the device has settled:
  # the launches of such a run are 3 - 6 % longer, getting shorter one by one); the counter passes need only a few
  TRACE_ARGS="--config $CFG --no-cpu-baseline --no-large-scan"
  [ "$CFG" = assoc ] && TRACE_ARGS="--config $CFG --no-cpu-baseline"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d ${P}_stats -- python3 bench.py $TRACE_ARGS > ${P}_stats.log 2>&1 || { echo "trace $CFG failed"; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d ${P}_sq -- python3 bench.py $ARGS > ${P}_sq.log 2>&1 || { echo "sq $CFG failed"; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d ${P}_fetch -- python3 bench.py $ARGS > ${P}_fetch.log 2>&1 || { echo "fetch $CFG failed"; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d ${P}_write -- python3 bench.py $ARGS > ${P}_write.log 2>&1 || { echo "write $CFG failed"; exit 1; }
  # the small summaries kept under profiles/ (the trace and counter directories themselves are scratch)
  python3 profiles/summarize.py stats ${P}_stats > gpurun_out/${TAG}_${CFG}_kernel_stats.csv
  python3 profiles/summarize.py counters ${P}_sq > gpurun_out/${TAG}_pmc_sq_${CFG}.json
  python3 profiles/summarize.py pmc ${P}_fetch ${P}_write > gpurun_out/${TAG}_pmc_traffic_${CFG}.json
  rm -rf ${P}_stats ${P}_fetch ${P}_write ${P}_sq
  echo "$CFG done"
done
