#!/bin/bash
# The round's measured evidence beyond tools/profile_round.sh: the loop's line with the driver's own arguments, the
# streaming sizes (2^23: a rank's slice of the sharded configuration at 8 GPUs; 2^26: the whole of it on one GPU),
# the association's all-pairs line, probes.  Writes gpurun_out/<tag>_*.
TAG=${1:-rXX}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_loop_steps20_warmup5.json 2> gpurun_out/${TAG}_e1.err || { echo "driver-style line failed"; exit 1; }
timeout -k 10 300 python3 bench.py --corr 8388608 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_loop_2p23.json 2> gpurun_out/${TAG}_e2.err || { echo "2^23 failed"; exit 1; }
timeout -k 10 400 python3 bench.py --corr 67108864 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_loop_2p26.json 2> gpurun_out/${TAG}_e3.err || { echo "2^26 failed"; exit 1; }
timeout -k 10 400 python3 bench.py --config assoc --method scan --steps 4 --warmup 1 > gpurun_out/${TAG}_bench_assoc_scan.json 2> gpurun_out/${TAG}_e4.err || { echo "assoc scan failed"; exit 1; }
timeout -k 10 300 python3 tools/resident_probe.py > gpurun_out/${TAG}_resident_probe.txt 2>&1 || { echo "resident probe failed"; exit 1; }
timeout -k 10 300 python3 tools/hold_probe.py > gpurun_out/${TAG}_hold_probe.txt 2>&1 || { echo "hold probe failed"; exit 1; }
timeout -k 10 300 python3 tools/assoc_shard_probe.py > gpurun_out/${TAG}_assoc_shard_probe.txt 2>&1 || { echo "assoc shard probe failed"; exit 1; }
timeout -k 10 300 python3 tools/sharded_overhead.py 2097152 8388608 33554432 > gpurun_out/${TAG}_sharded_overhead.txt 2>&1 || { echo "sharded overhead failed"; exit 1; }
echo "evidence done"
