"""Per-block iteration time of the sharded loop with a one-rank RCCL group (where does the slow start come from?)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from nmfgpu_amd.distributed import EngineShard, ShardedMU
import bench
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
V, W, H = bench.make_problem(0)
shard = EngineShard(V, W, H)
drv = ShardedMU(shard, total_columns=5000, rows=10000, force_collectives=True)
it = 1
out = []
for blk in range(30):
    t0 = time.perf_counter()
    drv.run(20, first_iteration=it, error_every=10); shard.synchronize()
    out.append(1e6 * (time.perf_counter() - t0) / 20); it += 20
print("us/iteration per block of 20:", " ".join(f"{v:.0f}" for v in out))
dist.destroy_process_group()
