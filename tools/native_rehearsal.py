"""Two ranks on ONE GPU (gloo process group): the RCCL id is made on rank 0, handed round and both
ranks call picp_comm_init together.  RCCL does not accept two ranks on one device, so the expected
outcome is an agreed refusal (init_native_exchange False on both ranks) followed by a correct run of
the generic exchange -- a rehearsal of the id hand-over and of the fallback, not of the collective."""
import importlib
import os
import sys
import time

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_main(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.synthetic import make_problem
    p = importlib.import_module("monocular-visual-odometry_amd")
    L = p._lib
    ctx = p.get_context(0)
    t0 = time.time()
    ok = p.sharded.init_native_exchange(ctx)
    print(f"rank {rank}: init_native_exchange -> {ok} in {time.time() - t0:.1f} s; "
          f"last error: {ctx.lib.picp_last_error().decode()!r}", flush=True)
    M = 100_000
    uv, pw, _ = make_problem(M, seed=9)
    lo, hi = p.sharded.shard_bounds(M, rank, world)
    cam = L.camera_struct([[180, 0, 320], [0, 180, 240], [0, 0, 1]], 640, 480)
    sched = L.schedule_struct(8, 10, 1000, 10, 2000, 64, 1, 64, True)
    stages = p.sharded.HipStages(ctx, ctx.soa(uv[lo:hi], pw[lo:hi]), np.eye(4), cam, sched)
    res, hist = p.sharded.run_sharded(stages, 8, disable_stop=True)
    print(f"rank {rank}: native={stages.native_exchange} iterations={res.num_iterations} "
          f"inliers={hist[:, L.H_NUM_INLIERS].astype(int).tolist()}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(rank_main, args=(2, 29533), nprocs=2, join=True)
