#!/bin/bash
# A/B of association builds: tools/ab_assoc.sh name ... (libraries ab/libpicp_<name>.so; "base" = the in-tree one)
cd "$(dirname "$0")/.."
for n in "$@"; do
  if [ $n = base ]; then unset PICP_LIB; else export PICP_LIB=$PWD/ab/libpicp_$n.so; fi
  timeout -k 10 120 python bench.py --config assoc --no-cpu-baseline --steps 40 > gpurun_out/ab_$n.json 2>/dev/null || exit 1
  echo $n $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_$n.json)
done
