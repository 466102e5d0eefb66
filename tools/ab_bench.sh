#!/bin/bash
# A/B timing of two builds of libpicp_hip.so on the same GPU box, alternating runs so that box and
# clock differences cancel:   tools/ab_bench.sh ab/libA.so ab/libB.so [rounds] [extra bench.py args]
A=$1; B=$2; ROUNDS=${3:-3}; shift 3 2>/dev/null
for r in $(seq $ROUNDS); do
  for v in "$A" "$B"; do
    PICP_LIB=$PWD/$v timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value'] / 1e9, 2), 'G corr-it/s   stage2', round(d['roofline']['mean_launch_ms'] * 1e3, 2), 'us   step', round(d['ms_per_step'], 4), 'ms')"
  done
done
