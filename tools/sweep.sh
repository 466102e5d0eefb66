#!/bin/bash
# sweeps the workgroups-per-CU knobs of the two scans on one box
for r in 1 2; do
for k1 in 1 2 4; do for k2 in 1 2; do
  PICP_STAGE1_WG_PER_CU=$k1 PICP_STAGE2_WG_PER_CU=$k2 timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('wg/cu stage1=$k1 stage2=$k2', round(d['value'] / 1e9, 2), 'G/s  stage2', round(d['roofline']['mean_launch_ms'] * 1e3, 2), 'us  step', round(d['ms_per_step'], 4))"
done; done; done
