#!/bin/bash
# Rehearsal on a box with ONE GPU of `python3 bench.py --gpus N` with no launcher around it (bench.py starts its
# own ranks): gloo process group, all ranks on device 0.  Writes under gpurun_out/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PICP_BENCH_SINGLE_DEVICE=1 PICP_BENCH_BACKEND=gloo
T=${1:-r03}
timeout -k 10 400 python3 bench.py --gpus 2 --corr 16777216 --steps 10 --warmup 2 > gpurun_out/${T}_selflaunch_loop2.json 2> gpurun_out/${T}_selflaunch_loop2.err || { echo "loop2 failed"; tail -20 gpurun_out/${T}_selflaunch_loop2.err; exit 1; }
timeout -k 10 400 python3 bench.py --gpus 4 --corr 16777216 --steps 10 --warmup 2 > gpurun_out/${T}_selflaunch_loop4.json 2> gpurun_out/${T}_selflaunch_loop4.err || { echo "loop4 failed"; tail -20 gpurun_out/${T}_selflaunch_loop4.err; exit 1; }
timeout -k 10 300 python3 bench.py --gpus 2 --config batched --steps 10 --warmup 2 > gpurun_out/${T}_selflaunch_batched2.json 2> gpurun_out/${T}_selflaunch_batched2.err || { echo "batched2 failed"; tail -20 gpurun_out/${T}_selflaunch_batched2.err; exit 1; }
timeout -k 10 300 python3 bench.py --gpus 2 --config assoc --steps 10 --warmup 2 > gpurun_out/${T}_selflaunch_assoc2.json 2> gpurun_out/${T}_selflaunch_assoc2.err || { echo "assoc2 failed"; tail -20 gpurun_out/${T}_selflaunch_assoc2.err; exit 1; }
echo "self-launch rehearsal done"
for f in loop2 loop4 batched2 assoc2; do python3 - "$f" "$T" <<'PY'
import json, sys
f, T = sys.argv[1], sys.argv[2]
d = json.loads(open(f"gpurun_out/{T}_selflaunch_{f}.json").read().strip().splitlines()[-1])
print(f, d["n_gpus"], f'{d["value"]:.4g}', d["unit"], d.get("replicas_identical"), (d.get("single_gpu_same_workload") or {}).get("value"), d["config"].get("collective"))
PY
done
