#!/bin/bash
# One-GPU rehearsal of the set-up failures a first run on a multi-GPU node could meet: each is injected once
# (PICP_INJECT_FAILURE, PICP_BENCH_FAIL_STEP) into `bench.py --gpus 2` and must end in the next fallback with a
# valid bench line.  gloo process group, both ranks on device 0.  Writes gpurun_out/<tag>_inject_*.json/.err.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${1:-r03}
export PICP_BENCH_SINGLE_DEVICE=1 PICP_BENCH_BACKEND=gloo PICP_COMM_INIT_TIMEOUT=20
ARGS="--gpus 2 --corr 4194304 --steps 5 --warmup 1 --no-single-gpu"
run() {  # name, then environment assignments
  local name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py $ARGS > gpurun_out/${T}_inject_${name}.json 2> gpurun_out/${T}_inject_${name}.err
  local rc=$?
  python3 - "$name" "$T" "$rc" <<'PY'
import json, sys
name, T, rc = sys.argv[1], sys.argv[2], int(sys.argv[3])
lines = [l for l in open(f"gpurun_out/{T}_inject_{name}.json").read().splitlines() if l.startswith("{")]
if rc != 0 or not lines:
    print(f"{name}: FAILED rc={rc}, no bench line"); sys.exit(1)
d = json.loads(lines[-1])
print(f'{name}: rc={rc} value={d["value"]:.4g} replicas_identical={d.get("replicas_identical")} collective={d["config"]["collective"]}')
PY
}
run none PICP_INJECT_FAILURE= || exit 1
run peer_access PICP_INJECT_FAILURE=peer_access || exit 1
run peer_create PICP_INJECT_FAILURE=peer_create || exit 1
run trial_bits PICP_INJECT_FAILURE=trial_bits || exit 1
run peer_uncached PICP_INJECT_FAILURE=peer_uncached || exit 1   # second attempt with fine-grained memory succeeds
run fail_step PICP_BENCH_FAIL_STEP=30 || exit 1
run comm_ready PICP_INJECT_FAILURE=peer_access,comm_ready PICP_BENCH_EXCHANGE=auto PICP_BENCH_NATIVE_ON_GLOO=1 || exit 1
run finegrained PICP_PEER_MEMORY=finegrained || exit 1
run comm_init PICP_INJECT_FAILURE=peer_access,comm_init PICP_BENCH_NATIVE_ON_GLOO=1 || exit 1
echo "failure rehearsal done"
