#!/bin/bash
# Rehearsal of bench.py with two ranks on a box with ONE GPU (gloo process group, both ranks on device 0), of the
# three configurations, plus the randomized path comparison.  Writes under gpurun_out/.
set -e
export PICP_BENCH_SINGLE_DEVICE=1 PICP_BENCH_BACKEND=gloo
R="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1"
timeout -k 10 200 python bench.py --config batched --steps 10 --warmup 2 > gpurun_out/s6b_batched1.json 2> gpurun_out/s6b_batched1.err
timeout -k 10 200 python bench.py --config assoc --steps 10 --warmup 2 > gpurun_out/s6b_assoc1.json 2> gpurun_out/s6b_assoc1.err
timeout -k 10 200 $R --master-port 29511 bench.py --gpus 2 --config batched --steps 10 --warmup 2 > gpurun_out/s6b_batched2.json 2> gpurun_out/s6b_batched2.err
timeout -k 10 300 $R --master-port 29512 bench.py --gpus 2 --config assoc --steps 10 --warmup 2 > gpurun_out/s6b_assoc2.json 2> gpurun_out/s6b_assoc2.err
timeout -k 10 300 $R --master-port 29513 bench.py --gpus 2 --corr 16777216 --steps 10 --warmup 2 > gpurun_out/s6b_loop2.json 2> gpurun_out/s6b_loop2.err
timeout -k 10 400 python tests/fuzz_paths.py 11 40 > gpurun_out/s6b_fuzz.log 2>&1
tail -2 gpurun_out/s6b_fuzz.log
