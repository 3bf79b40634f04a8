"""Is the sharded Python loop host-bound?  Times the enqueue phase (loop returns) and the total (after sync)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from nmfgpu_amd.distributed import EngineShard, ShardedMU
import bench
rccl = len(sys.argv) > 1 and sys.argv[1] == "rccl"
torch.cuda.set_device(0)
if rccl:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
V, W, H = bench.make_problem(0)
shard = EngineShard(V, W, H)
drv = ShardedMU(shard, total_columns=5000, rows=10000, force_collectives=rccl)
drv.run(30, first_iteration=1, error_every=10); shard.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    drv.run(200, first_iteration=31, error_every=10)
    t1 = time.perf_counter()
    shard.synchronize()
    t2 = time.perf_counter()
    print(f"{'rccl1' if rccl else 'plain'}: enqueue {1e6 * (t1 - t0) / 200:.1f} us/iteration, total {1e6 * (t2 - t0) / 200:.1f} us/iteration")
if rccl:
    dist.destroy_process_group()
