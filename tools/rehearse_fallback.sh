#!/bin/bash
# Two ranks on a box with ONE GPU: the last rank drops out of the in-kernel exchange in the middle of the timed
# steps; both ranks must agree on the failure and finish the measurement with the stage launches.
set -e
export PICP_BENCH_SINGLE_DEVICE=1 PICP_BENCH_BACKEND=gloo PICP_BENCH_FAIL_STEP=${PICP_BENCH_FAIL_STEP:-40}
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 \
    bench.py --gpus 2 --corr 4194304 --steps 10 --warmup 2 > gpurun_out/fallback2.json 2> gpurun_out/fallback2.err
tail -c 600 gpurun_out/fallback2.err
python -c "
import json; d=json.loads(open('gpurun_out/fallback2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['collective'])"
