"""GPU tests of the column-sharded path with the real HIP backend (EngineShard).

world = 1 runs in-process.  world = 2 runs two ranks that SHARE the box's single GPU and talk
over gloo (RCCL refuses two ranks on one device); the collective is therefore not RCCL, but
everything else -- the three-phase engine API, the exchange buffer layout, the error-term gather --
is the production code."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def F(a):
    return np.asfortranarray(a)


def _problem(m, n, r, seed=3):
    rng = np.random.default_rng(seed)
    V = F(rng.random((m, n)).astype(np.float32))
    W = F((1.0 - rng.random((m, r))).astype(np.float32))
    H = F((1.0 - rng.random((r, n))).astype(np.float32))
    return V, W, H


def _rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64))


@pytest.mark.parametrize("mode", ["replicated", "row_blocks"])
@pytest.mark.parametrize("alg,r,theta", [("mu", 16, 0.0), ("mu", 64, 0.0), ("nsnmf", 16, 0.5), ("nsnmf", 200, 0.3)])
def test_sharded_engine_world1_matches_oracle(alg, r, theta, mode):
    import torch  # noqa: F401  (device memory for the exchange buffer)
    from nmfgpu_amd.distributed import EngineShard, ShardedMU
    from oracle import oracle
    m, n, iters = 384, 256, 30
    V, W, H = _problem(m, n, r)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run(alg, V64, W64, H64, iters, theta=theta)
    shard = EngineShard(V, W, H, algorithm=alg, theta=theta)
    drv = ShardedMU(shard, total_columns=n, rows=m, mode=mode)
    drv.run(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wg, Hg = shard.factors()
    assert _rel(Wg, W64) < 2e-4 and _rel(Hg, H64) < 2e-4
    assert drv.frobenius == pytest.approx(ref["frobenius"], rel=1e-5)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, m, n, r, iters, out_dir, alg, theta, precision, mode="replicated"):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from nmfgpu_amd.distributed import EngineShard, ShardedMU
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # the box's hostname may not resolve: gloo would probe interfaces
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    V, W, H = _problem(m, n, r)
    per = n // world
    cols = slice(rank * per, (rank + 1) * per)
    shard = EngineShard(F(V[:, cols]), W, F(H[:, cols]), algorithm=alg, theta=theta, precision=precision, row_blocks=world if mode == "row_blocks" else 1)
    drv = ShardedMU(shard, total_columns=n, rows=m, mode=mode)
    drv.run(iters, first_iteration=1, error_every=10, last_iteration=iters)
    Wg, Hg = shard.factors()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), W=Wg, H=Hg, frob=drv.frobenius)
    dist.destroy_process_group()


@pytest.mark.parametrize("alg,r,theta,precision,tol,mode", [("mu", 16, 0.0, "native", 2e-4, "replicated"), ("nsnmf", 256, 0.5, "native", 2e-4, "replicated"),
                                                            ("nsnmf", 256, 0.5, "bf16", 2e-2, "replicated"), ("mu", 64, 0.0, "native", 2e-4, "row_blocks"),
                                                            ("nsnmf", 256, 0.5, "bf16", 2e-2, "row_blocks")])
def test_sharded_engine_two_ranks_on_one_gpu(tmp_path, alg, r, theta, precision, tol, mode):
    """Two ranks on the one GPU of the box, gloo for the exchange.  The last case is BASELINE config 4 in
    miniature: nsNMF, r = 256, bf16 operands, column shards."""
    import torch.multiprocessing as mp
    from oracle import oracle
    m, n, iters, world = 384, 512, 20, 2
    mp.spawn(_worker, args=(world, _free_port(), m, n, r, iters, str(tmp_path), alg, theta, precision, mode), nprocs=world, join=True)
    V, W, H = _problem(m, n, r)
    V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
    ref = oracle.run(alg, V64, W64, H64, iters, theta=theta)
    outs = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    per = n // world
    for k, o in enumerate(outs):
        assert _rel(o["W"], W64) < tol
        assert _rel(o["H"], H64[:, k * per:(k + 1) * per]) < tol
        assert float(o["frob"]) == pytest.approx(ref["frobenius"], rel=1e-5 if precision == "native" else 2e-3)
    assert np.array_equal(outs[0]["W"], outs[1]["W"])   # replicas bit-identical


def _rccl_worker(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from nmfgpu_amd.distributed import EngineShard, ShardedMU
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    try:      # an environment that cannot bring up RCCL at all is a skip, not a failure of the code under test
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        probe = torch.ones(8, device="cuda")
        dist.all_reduce(probe)
        torch.cuda.synchronize()
    except Exception as exc:      # noqa: BLE001
        with open(os.path.join(out_dir, "skip.txt"), "w") as f:
            f.write(repr(exc))
        return
    res = {}
    for alg, r, theta, precision in (("mu", 64, 0.0, "native"), ("nsnmf", 200, 0.4, "native"), ("nsnmf", 256, 0.5, "bf16")):
        V, W, H = _problem(640, 512, r)
        shard = EngineShard(V, W, H, algorithm=alg, theta=theta, precision=precision)
        drv = ShardedMU(shard, total_columns=512, rows=640, force_collectives=True)
        drv.run(20, first_iteration=1, error_every=10, last_iteration=20)
        Wg, Hg = shard.factors()
        res[f"{alg}{r}{precision}"] = (Wg, Hg, drv.frobenius)
    dist.barrier()
    np.savez(os.path.join(out_dir, "rccl.npz"), **{k + "_W": v[0] for k, v in res.items()}, **{k + "_H": v[1] for k, v in res.items()},
             **{k + "_f": np.array(v[2]) for k, v in res.items()})
    dist.destroy_process_group()


def test_sharded_engine_through_rccl_single_rank(tmp_path):
    """backend "nccl" (= RCCL) with a one-rank group on the box's GPU: the all-reduce of the exchange buffer and the
    all-gather of the error terms are issued for real (identities at world size 1), stream-ordered against the engine's
    kernels exactly as in a multi-GPU run.  What a 1-GPU box can prove about the RCCL path."""
    import torch.multiprocessing as mp
    from oracle import oracle
    mp.spawn(_rccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    if (tmp_path / "skip.txt").exists():
        pytest.skip("RCCL process group could not be created here: " + (tmp_path / "skip.txt").read_text())
    out = np.load(tmp_path / "rccl.npz")
    for alg, r, theta, precision, tol in (("mu", 64, 0.0, "native", 2e-4), ("nsnmf", 200, 0.4, "native", 2e-4), ("nsnmf", 256, 0.5, "bf16", 2e-2)):
        V, W, H = _problem(640, 512, r)
        V64, W64, H64 = (F(x.astype(np.float64)) for x in (V, W, H))
        ref = oracle.run(alg, V64, W64, H64, 20, theta=theta)
        key = f"{alg}{r}{precision}"
        assert _rel(out[key + "_W"], W64) < tol and _rel(out[key + "_H"], H64) < tol
        assert float(out[key + "_f"]) == pytest.approx(ref["frobenius"], rel=1e-5 if precision == "native" else 2e-3)
