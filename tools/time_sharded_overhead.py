"""Host cost of one sharded iteration (tiny problem: the GPU work is negligible, what remains is enqueue overhead)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from nmfgpu_amd.distributed import EngineShard, ShardedMU
rccl = len(sys.argv) > 1 and sys.argv[1] == "rccl"
torch.cuda.set_device(0)
if rccl:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29536")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
m, n, r = 2304, 2304, 64
rng = np.random.default_rng(1)
V = np.asfortranarray(rng.random((m, n)).astype(np.float32)); W = np.asfortranarray((1 - rng.random((m, r))).astype(np.float32)); H = np.asfortranarray((1 - rng.random((r, n))).astype(np.float32))
shard = EngineShard(V, W, H)
drv = ShardedMU(shard, total_columns=n, rows=m, force_collectives=rccl)
drv.run(100, first_iteration=1, error_every=0); shard.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); drv.run(500, first_iteration=1, error_every=0); t1 = time.perf_counter(); shard.synchronize(); t2 = time.perf_counter()
    print(f"{'rccl1' if rccl else 'plain'}: enqueue {1e6 * (t1 - t0) / 500:.1f} us/iteration, total {1e6 * (t2 - t0) / 500:.1f}")
if rccl:
    dist.destroy_process_group()
