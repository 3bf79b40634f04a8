"""bench.py -- NMF multiplicative-update iterations/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1: BASELINE config 2 -- dense V 10 000 x 5 000, r = 64, Lee-Seung MU (Frobenius), fp32, V resident
in HBM; one step = one MU iteration (error terms evaluated every 10th iteration, as in the reference's
loop, source/nmf/SingleGpuDispatcher.cpp:171-201).
N > 1: the SAME 10 000 x 5 000 problem column-sharded over the N GPUs (BASELINE north_star's 1/2/4/8 series; "scaling": "strong",
value = iterations/s of that problem; --scaling weak: one such matrix per GPU).  Launched by torch.distributed.run (one rank per
GPU) or by itself; two transports (team: one process, a rank thread per GPU, peer reads over xGMI; rccl: a process per GPU, RCCL's
C API), every phase under a deadline -- see "N > 1" below.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M, N_COLS, R = 10000, 5000, 64
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: "Peak FP32 (matrix) 157.3 TFLOPS spec"
PEAK_HBM_GBS = 8000.0           # same guide: HBM3E ~8 TB/s spec (6.29 TB/s measured copy rate)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: "Peak BF16/FP16 MFMA ~2.5 PF dense" (the 5 PF figure counts 2:1 sparsity)

# --workload c4 (NOT the default and not the BASELINE metric line): one column shard of BASELINE config 4 per GPU
C4 = {"rows": 50000, "columns_per_gpu": 6250, "features": 256, "theta": 0.5}
# --workload c3: BASELINE configs[2], sparse CSR V, KL-divergence MU (SURVEY.md section 8d's synthetic input)
C3 = {"rows": 100000, "columns": 20000, "features": 128, "row_nnz_mean": 200}
# --workload c5 / c5-gdcls: BASELINE configs[4], the constrained variants at config 2's shape; parameter values of the
# reference's example (example/main.cpp:119-126)
C5 = {"ahcls": dict(lambda_w=0.01, lambda_h=0.01, alpha_w=0.01, alpha_h=0.01), "gdcls": dict(lam=0.01)}
PEAK_L2_GATHER_GBS = 18000.0    # MI355X_MICROARCH.md "Indexed rows": 16.8-18.8 TB/s chip-wide for rows served by the XCD's L2


def measured_traffic(name: str, source: str):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (tools/pmc_traffic.py writes
    profiles/traffic_<name>.json together with the sha256 of the kernel source it measured).  Counters cannot be collected
    inside this process, so the figure is only reported while the kernel source is still the one that was measured."""
    import hashlib
    path = os.path.join(ROOT, "profiles", f"traffic_{name}.json")
    if not os.path.exists(path):
        return None, "no PMC measurement committed"
    d = json.load(open(path))
    src = os.path.join(ROOT, source)
    if d.get("source_sha256") != (hashlib.sha256(open(src, "rb").read()).hexdigest() if os.path.exists(src) else None):
        return None, f"stale: {source} changed since profiles/traffic_{name}.json was measured"
    return d.get("hbm_bytes_per_launch"), f"profiles/traffic_{name}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, kernel source hash matches)"


def make_problem(shard: int, m: int = M, n: int = N_COLS, r: int = R):
    """V = U[0,1) fp32 from mt19937(1 + shard); W0, H0 = U(0,1] from seeds 2, 3."""
    rs = np.random.RandomState(1 + shard)
    V = np.empty((m, n), dtype=np.float32, order="F")
    for j0 in range(0, n, 1000):       # column blocks: the same stream as one (n, m) draw, without the fp64 temporary
        V[:, j0:j0 + 1000] = rs.random_sample((min(1000, n - j0), m)).astype(np.float32).T
    W = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((r, m))).astype(np.float32).T)
    H = np.asfortranarray((1.0 - np.random.RandomState(3 + 1000 * shard).random_sample((n, r))).astype(np.float32).T)
    return V, W, H


def make_sparse_problem(m: int = C3["rows"], n: int = C3["columns"], r: int = C3["features"], mean: int = C3["row_nnz_mean"], seed: int = 1):
    """BASELINE configs[2]: CSR int32, per-row count ~ Poisson(mean) clipped to >= 1, columns
    uniform (duplicates inside a row removed, so the count is a hair below the draw), values U{1..5}, base 0;
    W0, H0 = U(0,1].  Returns (values f32, rowPtr, columnIndices, W0, H0)."""
    rng = np.random.default_rng(seed)
    counts = np.maximum(rng.poisson(mean, size=m), 1).astype(np.int64)
    rows = np.repeat(np.arange(m, dtype=np.int64), counts)
    cols = rng.integers(0, n, size=rows.size, dtype=np.int64)
    key = np.unique(rows * n + cols)                     # sorted by (row, column), duplicates dropped
    rows = (key // n).astype(np.int64)
    idx = (key % n).astype(np.int32)
    ptr = np.zeros(m + 1, dtype=np.int64)
    np.add.at(ptr, rows + 1, 1)
    ptr = np.cumsum(ptr).astype(np.int32)
    val = rng.integers(1, 6, size=idx.size).astype(np.float32)
    W = np.asfortranarray((1.0 - rng.random((r, m))).astype(np.float32).T)
    H = np.asfortranarray((1.0 - rng.random((n, r))).astype(np.float32).T)
    return val, ptr, idx, W, H


def whole_iteration(roofline, step_s, floor_bytes=None, floor_flops=None, basis=""):
    """roofline.whole_iteration_frac: the fraction of the step time that the iteration's algorithmic traffic (or arithmetic) would take
    at the peak of the roofline's bound -- the number that tracks `value`, next to the dominant kernel's own fraction."""
    if roofline is None or step_s <= 0:
        return roofline
    if floor_bytes is not None:
        floor_s = floor_bytes / (roofline["peak"] * 1e9)
    else:
        floor_s = floor_flops / (roofline["peak"] * 1e12)
    roofline["whole_iteration_frac"] = floor_s / step_s
    roofline["whole_iteration_basis"] = basis
    return roofline


_ENGINE_STREAM = []


def engine_stream(torch) -> int:
    """A dedicated stream for the engine (kept alive for the process) instead of the legacy default stream; an A/B on one box showed no
    difference at config 2 (98.6 - 100.1 us either way).  The timed region is bracketed by torch.cuda.synchronize(), which waits for
    every stream of the device."""
    if os.environ.get("NMFAMD_BENCH_DEFAULT_STREAM"):      # (A/B switch)
        return torch.cuda.current_stream().cuda_stream
    if not _ENGINE_STREAM:
        _ENGINE_STREAM.append(torch.cuda.Stream())
    return _ENGINE_STREAM[0].cuda_stream


# Untimed set-up iterations before the W warm-up steps of a single-GPU line (config 2's shape; a third of it for config 4's 0.44 ms iteration, a tenth for config 3's 1.3 ms):
# ~25 ms of the hot path, after which the factors go back to W0, H0.  The line reports it as config.setup_iterations.
SETUP_ITERATIONS = 240
# Untimed replay behind the timed steps of a single-GPU line: at least this many further iterations, REPLAY_SAMPLES of them with their product launches timed by the
# launches' own start / stop events (roofline.avg_launch_us)
REPLAY_ITERATIONS, REPLAY_SAMPLES = 60, 12
TIMED_BY_REPLAY = "sampled in an untimed replay right behind the timed steps (same engine, same stream, iterations continue); the timed region carries no events"


def replay_iterations(K: int, c4: bool) -> int:
    return max(REPLAY_ITERATIONS // (3 if c4 else 1), min(K, 200))


def host_cpu_info() -> dict:
    """What the CPU legs run on, probed ("the GPU box is probed, never assumed"): the cores this process may use -- its affinity mask, cut to the
    cgroup's CPU quota where one is set (more threads than the quota only queue behind each other) -- and the CPU model."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:          # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max" and float(per) > 0:
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:      # cgroup v1
                q, per = float(f.read()), float(g.read())
                if q > 0 and per > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    available = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    return {"cores_available": available, "affinity_cores": affinity, "cgroup_quota_cores": quota, "cpu_model": model}


def set_cpu_threads(requested: int) -> None:
    """Before the oracle is loaded: its OpenMP team takes every available core (host_cpu_info) unless --cpu-threads caps it."""
    n = host_cpu_info()["cores_available"] if requested <= 0 else requested
    os.environ["NMF_ORACLE_MAX_THREADS"] = str(n)
    os.environ["OMP_NUM_THREADS"] = str(n)


def cpu_baseline(V, W, H, budget_s: float = 20.0, algorithm: str = "mu", **kw):
    """The oracle (our CPU port of the reference's iteration) timed on this box's host cores.  A run's one-off work (the sorted
    tr(V^T V) vector of allocateMemory, workspace allocation) is timed by a zero-iteration run and taken out: the metric counts
    iterations, on the GPU side as well."""
    from oracle import oracle
    Wc, Hc = W.copy(order="F"), H.copy(order="F")
    t0 = time.perf_counter()
    oracle.run(algorithm, V, Wc, Hc, 1, **kw)
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    oracle.run(algorithm, V, Wc, Hc, 0, **kw)
    setup = time.perf_counter() - t0
    iters = int(max(2, min(100, budget_s // max(first, 1e-3))))
    t0 = time.perf_counter()
    oracle.run(algorithm, V, Wc, Hc, iters, **kw)
    dt = max(time.perf_counter() - t0 - setup, 1e-9)
    return {"value": iters / dt, "unit": "iterations/s", "cores": oracle.num_threads(), **host_cpu_info(), "kind": "port",
            "sample": f"{iters} {algorithm.upper()} iterations of the full {V.shape[0]}x{V.shape[1]} r={W.shape[1]} {'fp32' if V.dtype == np.float32 else 'fp64'} problem (oracle/nmf_oracle.c, OpenMP; "
                      + ("products by oracle/sgemm_avx2.h; " if V.dtype == np.float32 else "") + f"{setup * 1e3:.0f} ms of per-run setup timed apart and excluded)"}


def cpu_baseline_blas(V, W, H, threads: int, budget_s: float = 6.0):
    """Second, stronger CPU line: the same MU iteration written with numpy on the bundled OpenBLAS
    (no oracle, no reference code), fp32, `threads` BLAS threads.  A reported figure next to cpu_baseline, nothing more."""
    try:
        from threadpoolctl import threadpool_limits
    except Exception:      # noqa: BLE001
        threadpool_limits = None
    eps = np.float32(np.finfo(np.float32).eps)
    Wc, Hc = np.ascontiguousarray(W), np.ascontiguousarray(H)
    Vc, Vt = np.ascontiguousarray(V), np.ascontiguousarray(V.T)

    def iteration():
        nonlocal Wc, Hc
        G = Wc.T @ Wc
        Hc *= (Wc.T @ Vc) / (G @ Hc + eps)
        Wc *= (Vt.T @ Hc.T) / (Wc @ (Hc @ Hc.T) + eps)
        nrm = np.sqrt((Wc * Wc).sum(axis=0)); nrm[nrm == 0] = 1
        Wc /= nrm

    ctx = threadpool_limits(limits=threads) if threadpool_limits else None
    try:
        t0 = time.perf_counter(); iteration(); first = time.perf_counter() - t0
        iters = int(max(1, min(40, budget_s // max(first, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(iters):
            iteration()
        dt = time.perf_counter() - t0
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    return {"value": iters / dt, "unit": "iterations/s", "cores": threads, "kind": "numpy + OpenBLAS, fp32",
            "sample": f"{iters} MU iterations of the full 10000x5000 r=64 problem"}


# ---- N > 1: launching, deadlines, phases -----------------------------------------------------------------------------------------------------------
# Two transports carry the per-iteration exchange of a column-sharded run:
#   team  -- ONE process, N rank threads, one per GPU (include/nmfgpu_amd.h, nmfamd_local_group_*): every rank's W update reads the peers' exchange panels
#            where they lie (xGMI peer access), no reduction kernel, one rendezvous per iteration.  What nmfgpu::compute runs with Parameter "numGpus".
#   rccl  -- one process per GPU (torch.distributed.run), RCCL through its C API on the engine's stream; torch.distributed (gloo) only carries the unique id,
#            the barriers and the MAX of the timings.
# `--transport auto` (default): team first, rccl when the team could not be set up (devices that cannot map each other's memory, a failed rank).  Every
# phase runs under ONE deadline (NMFAMD_BENCH_DEADLINE seconds, default 480): a run that passes it is killed -- child ranks included, which are fresh
# processes, never a re-exec of a process that touched the GPU -- and bench.py exits non-zero naming the phase that hung.
_PHASE = {"name": "start", "file": os.environ.get("NMFAMD_BENCH_PHASE_FILE")}


def deadline_seconds() -> float:
    return float(os.environ.get("NMFAMD_BENCH_DEADLINE", "480"))


def phase(name: str):
    """Where this process is, for the message of a deadline kill (kept in a file the launcher reads after it killed the process)."""
    _PHASE["name"] = name
    if _PHASE["file"]:
        try:
            with open(_PHASE["file"], "w") as f:
                f.write(name)
        except OSError:
            pass


def start_watchdog(what: str):
    """A rank under torch.distributed.run cannot be killed by bench.py's own launcher (there is none): it ends itself, non-zero, when the deadline passes --
    the elastic agent then ends the other ranks."""
    import threading
    t_end = time.monotonic() + deadline_seconds()

    def watch():
        while time.monotonic() < t_end:
            time.sleep(0.5)
        print(f"bench.py: {what} passed the deadline of {deadline_seconds():.0f} s in phase '{_PHASE['name']}': giving up", file=sys.stderr, flush=True)
        os._exit(3)
    threading.Thread(target=watch, daemon=True).start()


def run_child(cmd, env, what: str, timeout: float):
    """A child process (group) under a deadline.  Returns (returncode, stdout, stderr, hung_phase); the whole process group is killed when the time is up."""
    import signal
    import subprocess
    import tempfile
    fd, phase_file = tempfile.mkstemp(prefix="nmfamd_bench_phase_")
    os.close(fd)
    env = dict(env, NMFAMD_BENCH_PHASE_FILE=phase_file)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    hung = None
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            p.kill()
        out, err = p.communicate()
        try:
            hung = open(phase_file).read().strip() or "start"
        except OSError:
            hung = "unknown"
        print(f"bench.py: {what} passed its deadline of {timeout:.0f} s in phase '{hung}': killed", file=sys.stderr, flush=True)
    finally:
        try:
            os.unlink(phase_file)
        except OSError:
            pass
    return (p.returncode if hung is None else 3), out, err, hung


def worker_argv(args, role: str):
    argv = [sys.executable, os.path.abspath(__file__), role, "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--workload", args.workload,
            "--scaling", args.scaling, "--shard-mode", str(args.shard_mode), "--event-stride", str(args.event_stride), "--setup-iterations", str(args.setup_iterations)]
    for flag, on in (("--no-cpu-baseline", args.no_cpu_baseline), ("--no-kernel-events", args.no_kernel_events), ("--allow-shared-device", args.allow_shared_device)):
        if on:
            argv.append(flag)
    return argv


def clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                              "TORCHELASTIC_RUN_ID", "NMFAMD_BENCH_PHASE_FILE")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def launch_team(args, timeout: float):
    """The in-library team as a child process (one process, N rank threads).  Returns (ok, json_line or None, returncode)."""
    rc, out, err, hung = run_child(worker_argv(args, "--team-worker"), clean_env(), f"the {args.gpus}-rank team", timeout)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if rc == 0 and len(lines) == 1:
        return True, lines[0], 0
    sys.stderr.write(err[-4000:])
    return False, None, rc


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun.  BEFORE anything in this process touches the GPU (counting devices does not): fewer than N devices -> refuse
    (non-zero exit, nothing on stdout) rather than print a line measured on fewer GPUs than it claims, unless --allow-shared-device asks for a rehearsal.  Then
    the team (one child process), and if that cannot be set up the rank processes under torch.distributed.run on the loopback interface -- each under the deadline."""
    import socket
    import torch
    devices = torch.cuda.device_count()
    if devices < args.gpus:
        if not args.allow_shared_device:
            print(f"bench.py: --gpus {args.gpus} asked for, {devices} HIP device(s) visible: not running (a line measured on fewer GPUs than it names "
                  f"would be wrong).  --allow-shared-device rehearses {args.gpus} ranks on the devices there are.", file=sys.stderr, flush=True)
            raise SystemExit(2)
        if devices < 1:
            print("bench.py needs a HIP device: the engine has no CPU fallback", file=sys.stderr, flush=True)
            raise SystemExit(2)
    t_end = time.monotonic() + deadline_seconds()
    if args.transport in ("auto", "team") or devices < args.gpus:          # (ranks that share a device: the team is the only transport -- RCCL wants a device per rank)
        ok, line, rc = launch_team(args, max(5.0, t_end - time.monotonic()))
        if ok:
            print(line, flush=True)
            raise SystemExit(0)
        if args.transport == "team" or devices < args.gpus or rc == 3:
            raise SystemExit(rc or 1)
        print("bench.py: the team could not be set up; trying one process per GPU over RCCL", file=sys.stderr, flush=True)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:]]
    if "--transport" not in argv:
        argv += ["--transport", "rccl"]
    else:
        argv[argv.index("--transport") + 1] = "rccl"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    rc, out, err, hung = run_child(cmd, clean_env(), f"the {args.gpus} RCCL ranks", max(5.0, t_end - time.monotonic()))
    sys.stdout.write(out)
    sys.stderr.write(err[-6000:])
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the cpu_baseline legs; 0 (default) = every core available to this process (affinity mask, cgroup quota)")
    ap.add_argument("--setup-iterations", type=int, default=-1,
                    help="untimed iterations of the hot path BEFORE the --warmup steps (first launches of every kernel, the device's ramp from idle; the factors go back to W0, H0 "
                         "afterwards).  -1 (default): 240 at config 2's shape (scaled for the slower workloads), as in round 4 -- the line reports what ran in config.setup_iterations; "
                         "0: the timed steps follow the driver's warm-up directly")
    ap.add_argument("--sharded", action="store_true", help="N = 1 only: drive the three-phase sharded API from Python (no collective) instead of the fused loop")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket any factor-product launch with HIP events")
    ap.add_argument("--event-stride", type=int, default=0, help="(kept for old command lines; unused since round 5: the launch samples are taken in an untimed replay behind the timed steps)")
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c5", "c5-gdcls", "c2-f64", "example"], default="c2",
                    help="c2 (default) = BASELINE configs[1], the metric line; c3 = configs[2] (sparse CSR, KL-divergence MU, r=128); "
                         "c4 = one configs[3] column shard per GPU (nsNMF, r=256, bf16 operands); c5 / c5-gdcls = configs[4] (AHCLS / GDCLS at config 2's shape); "
                         "c2-f64 = config 2's shape in DOUBLE precision (the instantiation the reference's own callers run); example = the reference's example program "
                         "(example/main.cpp:31-33,119-126: 4096 x 165, r = 158, nsNMF theta 0.5, double)")
    ap.add_argument("--scaling", choices=["auto", "weak", "strong"], default="auto",
                    help="N > 1, c2: strong (what auto means there) = the ONE 10000x5000 problem of configs[1] column-sharded over the N GPUs -- BASELINE north_star's "
                         "1/2/4/8-GPU series, value in iterations/s of that problem; weak = one 10000x5000 column shard per GPU.  c4 is one shard per GPU (weak) by definition")
    ap.add_argument("--transport", choices=["auto", "team", "rccl"], default="auto",
                    help="N > 1: team = one process, one rank thread per GPU, the W update reads the peers' exchange panels in place (xGMI peer access); rccl = one process per "
                         "GPU, RCCL through its C API; auto (default) = team, and rccl when the team cannot be set up")
    ap.add_argument("--shard-mode", type=int, choices=[-1, 0, 1], default=-1,
                    help="W step of the sharded loop: 0 reduce-scatter / all-gather by row blocks of W, 1 replicated update on the summed exchange buffer, "
                         "-1 (default) by message size: row blocks when the m x r panel is 8 MB or more (config 4)")
    ap.add_argument("--allow-shared-device", action="store_true",
                    help="--gpus N with fewer than N HIP devices visible: rehearse with rank threads sharing devices instead of refusing; the line says so in config.parallelism")
    ap.add_argument("--team-worker", action="store_true", help=argparse.SUPPRESS)      # (internal: the child process that runs the N rank threads)
    args = ap.parse_args()
    if args.event_stride <= 0:
        args.event_stride = max(10, args.steps // 5)
    global SETUP_ITERATIONS
    if args.setup_iterations >= 0:
        SETUP_ITERATIONS = args.setup_iterations
    set_cpu_threads(args.cpu_threads)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if os.environ.get("NMFAMD_BENCH_DUMP_AFTER"):
        # debugging aid: where is every thread after N seconds (a rank that waits for its peers says nothing otherwise)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["NMFAMD_BENCH_DUMP_AFTER"]), exit=False)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1:
        if args.workload not in ("c2", "c4"):
            raise SystemExit(f"--workload {args.workload} is a single-GPU workload (BASELINE: 1 x MI355X)")
        if args.team_worker:
            return team_worker(args)
        if world > 1:
            return rccl_ranks(args)            # under torch.distributed.run (the driver's launch, or self_launch's)
        return self_launch(args)
    if args.workload == "c3":
        return main_c3(args)
    if args.workload == "c4":
        return main_c4(args)
    if args.workload in ("c2-f64", "example"):
        return main_f64(args)
    algorithm, alg_kw = "mu", {}
    if args.workload.startswith("c5"):
        algorithm = "gdcls" if args.workload.endswith("gdcls") else "ahcls"
        alg_kw = C5[algorithm]

    import torch
    import nmfgpu_amd as na

    rank = 0
    if not torch.cuda.is_available() or na.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    torch.cuda.set_device(0)

    V, W, H = make_problem(rank)
    K, Wm = args.steps, args.warmup

    def barrier():
        torch.cuda.synchronize()

    kernel_ms, kernel_launches, pair_overhead_ms = 0.0, 0, 0.0
    form_ms, form_launches, resident_images = (0.0, 0.0), (0, 0), 2
    world = 1
    if not args.sharded:
        eng = na.Engine(M, N_COLS, R, algorithm, dtype=np.float32, stream=engine_stream(torch), **alg_kw)
        eng.upload(V)
        eng.set_factors(W, H)
        # First leg (round 6): the K steps right behind the driver's W warm-up steps and NOTHING else -- what `--warmup W` means taken literally; reported as
        # value_driver_warmup_only / ms_per_step_driver_warmup_only beside the line's value, so that rounds stay comparable whatever the set-up phase below does.
        cold_elapsed = None
        if SETUP_ITERATIONS > 0:
            eng.iterate(Wm, first_iteration=1, error_every=10)
            eng.synchronize()
            barrier()
            t0 = time.perf_counter()
            eng.iterate(K, first_iteration=Wm + 1, error_every=10)
            barrier()
            cold_elapsed = time.perf_counter() - t0
            eng.set_factors(W, H)
        # set-up, not steps (as in the multi-GPU paths below): the first launches of every kernel and the device's ramp from idle -- measured, iterations 26-75 of a
        # process run at 101 us, 76-100 at 96, from ~125 on at 91-92 (profiles/r04_warmup_timeline.txt; the same ramp follows 2 s of idling) -- then back to W0, H0
        if SETUP_ITERATIONS > 0:
            eng.iterate(SETUP_ITERATIONS, first_iteration=1, error_every=10)
        eng.synchronize()
        eng.set_factors(W, H)
        eng.iterate(Wm, first_iteration=1, error_every=10)
        eng.synchronize()
        barrier()
        t0 = time.perf_counter()
        eng.iterate(K, first_iteration=Wm + 1, error_every=10)
        barrier()
        elapsed = time.perf_counter() - t0
        if not args.no_kernel_events:
            # the launch samples of `roofline` come from an UNTIMED replay right behind the timed steps (iterations Wm + K + 1 ..., same stream, same engine, the device
            # still warm): a sampled iteration costs the stream ~23 us of waits around its two timed launches, which round 4 paid inside the timed region
            replay = replay_iterations(K, False)
            eng.kernel_timing(max(1, replay // REPLAY_SAMPLES))
            eng.iterate(replay, first_iteration=Wm + K + 1, error_every=10)
            eng.synchronize()
            kernel_ms, kernel_launches, pair_overhead_ms, form_ms, form_launches = eng.kernel_timing_read3()
            eng.kernel_timing(0)
        frob = eng.frobenius
        product_kernel = eng.geometry()["product_kernel"]
        resident_images = eng.geometry()["resident_images"]
        parallelism = "single GPU"
    else:
        cold_elapsed = None
        from nmfgpu_amd.distributed import EngineShard, ShardedMU
        shard = EngineShard(V, W, H)
        drv = ShardedMU(shard, total_columns=N_COLS, rows=M)
        drv.run(30, first_iteration=1, error_every=10)      # set-up (first launches), then back to W0, H0
        shard.synchronize()
        shard.engine.set_factors(W, H)
        drv.run(Wm, first_iteration=1, error_every=10)
        shard.synchronize()
        barrier()
        t0 = time.perf_counter()
        drv.run(K, first_iteration=Wm + 1, error_every=10)
        barrier()
        elapsed = time.perf_counter() - t0
        if not args.no_kernel_events:
            replay = replay_iterations(K, False)
            shard.engine.kernel_timing(max(1, replay // REPLAY_SAMPLES))
            drv.run(replay, first_iteration=Wm + K + 1, error_every=10)
            shard.synchronize()
            kernel_ms, kernel_launches, pair_overhead_ms = shard.engine.kernel_timing_read2()
        frob = drv.frobenius
        product_kernel = shard.engine.geometry()["product_kernel"]
        parallelism = "single GPU, three-phase sharded API driven from Python (team of one)"

    if rank == 0:
        flops_per_launch = 2.0 * M * N_COLS * R               # one product against V (algorithmic, unpadded)
        bytes_per_launch = 4.0 * M * N_COLS                   # the fp32 image of V (or V^T) one product streams
        roofline = roofline_mfma = None
        if kernel_launches > 0:
            # The split-operand product's launches carry their own start / stop events (hipExtLaunchKernel: the dispatch's timestamps, as rocprofv3 reads
            # them); the other product kernels are bracketed by two recorded events, which over-report a launch by a few us (an EMPTY pair on the idle
            # stream reports idle_event_pair_us).  No correction is applied either way.
            avg_s = kernel_ms / 1e3 / kernel_launches
            common = {"avg_launch_us": avg_s * 1e6, "idle_event_pair_us": pair_overhead_ms * 1e3, "launches": kernel_launches,
                      "timed_by": ("start/stop events of the launch itself (hipExtLaunchKernel), " if product_kernel == 2 else "an event recorded before and one after the launch, ") + TIMED_BY_REPLAY,
                      "flops_per_launch": flops_per_launch, "bytes_per_launch": bytes_per_launch}
            if product_kernel == 2:
                # fp32 product on the bf16 matrix pipe (operands split exactly into 3 bf16 terms): six bf16 MFMAs replace
                # sixteen fp32-MFMA-equivalents, and the kernel is bound by streaming the fp32 image of V from HBM
                traffic, traffic_source = measured_traffic("factor_product_x3", "nmfgpu_amd/csrc/kernels_x3.hip")
                # `achieved` is the rate at which the kernel consumes the fp32 image of V (algorithmic bytes / launch time): with
                # ONE resident image (config 2) part of it is served by the 256 MiB memory-side cache, so this is an effective
                # bandwidth against the HBM peak, not a DRAM measurement; `traffic` is what leaves L2 (FETCH_SIZE x 2 + WRITE_SIZE)
                roofline = {"bound": "hbm", "achieved": bytes_per_launch / avg_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": bytes_per_launch / avg_s / 1e9 / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_source,
                            "achieved_is": "effective bandwidth: algorithmic bytes of V per launch / launch time (memory-side cache hits included)",
                            "kernel": "k_factor_product_x3", "fp32_equivalent_tflops": flops_per_launch / avg_s / 1e12, **common}
                # "hbm" is the nearest of the contract's two labels, not a saturated resource: ONE image of V (207 MB) is resident and the second product of
                # an iteration finds most of it in the 256 MiB memory-side cache; the matrix pipe is the other candidate -- see roofline_mfma below
                roofline["streams_from"] = ("HBM + the 256 MiB memory-side cache (one resident fp32 image of V: the product that runs second in an iteration reads most of it from that cache)"
                                            if resident_images == 1 else "HBM (two resident images of V, each streamed along its output index)")
                # the two launches of an iteration are two FORMS of the kernel on the same image: W^T V reads it along the reduction index
                # (y-tiled form, staged through LDS), V H^T along its output index (x-tiled form)
                if form_launches[0] > 0 and form_launches[1] > 0:
                    forms = {}
                    for key, what, ms, cnt in (("wt_v", "W^T V: y-tiled form (reads the one image along the reduction index, straight to registers)" if resident_images == 1 else "W^T V: x-tiled form on the image of V^T", form_ms[0], form_launches[0]),
                                               ("v_ht", "V H^T: x-tiled form (reads the image along its output index)", form_ms[1], form_launches[1])):
                        a = ms / 1e3 / cnt
                        forms[key] = {"form": what, "avg_launch_us": a * 1e6, "launches": cnt, "achieved": bytes_per_launch / a / 1e9, "frac": bytes_per_launch / a / 1e9 / PEAK_HBM_GBS}
                    roofline["forms"] = forms
                # the same launches against the matrix pipe: six bf16 MFMAs per fp32 product term are what the kernel issues (12 mnr flops per launch); the share of
                # the pipe's cycles they occupy, measured with SQ counters: profiles/r05_pmc_sq.md (55 - 57 % busy; round 3's kernel: 43 - 48 %)
                mfma_tflops = 6.0 * flops_per_launch / avg_s / 1e12
                roofline_mfma = {"bound": "mfma", "achieved": mfma_tflops, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": mfma_tflops / PEAK_BF16_MFMA_TFLOPS,
                                 "kernel": "k_factor_product_x3", "flops_per_launch": 6.0 * flops_per_launch,
                                 "achieved_is": "bf16 MFMA work issued (six exact cross terms per fp32 product) / launch time, against the dense bf16 peak",
                                 "pipe_busy_measured": "55-57 % of cycles (SQ_VALU_MFMA_BUSY_CYCLES, profiles/r05_pmc_sq.md)"}
            else:
                achieved = flops_per_launch / avg_s / 1e12
                traffic, traffic_source = measured_traffic("factor_product", "nmfgpu_amd/csrc/kernels.hip")
                roofline = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_source, "kernel": "k_factor_product_f32", **common}
        names = {"mu": "MU", "ahcls": "AHCLS", "gdcls": "GDCLS"}
        out = {
            "metric": f"NMF {names[algorithm]} iterations/sec, dense 10kx5k r=64",
            "value": world * K / elapsed,
            "unit": "iterations/s" if world == 1 else "shard-iterations/s (one 10000x5000 column shard per GPU)",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": elapsed / K * 1e3,
            # the same K steps timed right behind the driver's W warm-up steps, before the set-up phase ran (same process, same engine; null when --setup-iterations 0
            # makes `value` itself that figure)
            "value_driver_warmup_only": (world * K / cold_elapsed) if cold_elapsed else None,
            "ms_per_step_driver_warmup_only": (cold_elapsed / K * 1e3) if cold_elapsed else None,
            # (N = 1 is the first point of north_star's 1/2/4/8 series of the ONE 10000 x 5000 problem, which `--gpus N` column-shards: strong scaling)
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: dense random V 10000x5000, r=64, MU Frobenius, fp32" if algorithm == "mu" else
                                   f"configs[4]: {names[algorithm]} at dense random V 10000x5000, r=64, fp32, parameters {alg_kw}",
                       "rows": M, "columns": N_COLS, "features": R, "error_every": 10, "setup_iterations": 0 if args.sharded else SETUP_ITERATIONS, "parallelism": parallelism,
                       "arithmetic": ("fp32 operands and fp32 accumulation; the two big products run on the bf16 matrix pipe with every operand "
                                      "split EXACTLY into three bf16 terms (six cross products kept, dropped terms <= 2^-23 relative): measured "
                                      "error against fp64 equals the native fp32 MFMA kernel's (tests/test_gpu_parity.py)") if product_kernel == 2
                                     else "fp32 MFMA instructions"},
            "frobenius_last": frob,
            # MU 4 m n r + 4 r^2 (m + n); the constrained variants add ~6 r^2 (m + n) for the two solve chains
            "iter_flops": 4.0 * M * N_COLS * R + (4.0 if algorithm == "mu" else 10.0) * R * R * (M + N_COLS),
            "achieved_tflops_whole_iteration": (4.0 * M * N_COLS * R + (4.0 if algorithm == "mu" else 10.0) * R * R * (M + N_COLS)) * (K / elapsed) / 1e12,
            "roofline": (whole_iteration(roofline, elapsed / K, floor_bytes=2.0 * bytes_per_launch, basis="two passes over the fp32 image of V per iteration (2 x 200 MB) at the HBM peak")
                         if roofline is not None and roofline["bound"] == "hbm" else
                         whole_iteration(roofline, elapsed / K, floor_flops=2.0 * flops_per_launch, basis="the two products against V (2 x 6.4 GFLOP) at the fp32 MFMA peak")),
        }
        if roofline_mfma is not None:
            out["roofline_mfma"] = roofline_mfma
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(V, W, H, algorithm=algorithm, **alg_kw)
            if algorithm == "mu":
                out["cpu_baseline_blas"] = cpu_baseline_blas(V, W, H, out["cpu_baseline"]["cores"])
        print(json.dumps(out), flush=True)


def multi_problem(args, world: int, rank: int):
    """This rank's share of an N-GPU run.  c2, strong (default; BASELINE north_star's 1/2/4/8 series): the ONE 10000 x 5000 matrix of configs[1], columns dealt to the
    ranks; c2, weak: one 10000 x 5000 matrix per rank; c4: one 50000 x 6250 column shard of configs[3] per rank (its stated cut: 8 shards)."""
    import nmfgpu_amd as na
    c4 = args.workload == "c4"
    strong = (not c4) and args.scaling != "weak"
    if args.scaling == "strong" and c4:
        raise SystemExit("--scaling strong is defined for configs[1] (the north_star's 1/2/4/8-GPU series); configs[3] is one shard per GPU")
    rows, cols, feats = (C4["rows"], C4["columns_per_gpu"], C4["features"]) if c4 else (M, N_COLS, R)
    alg, alg_kw = ("nsnmf", dict(theta=C4["theta"], precision="bf16")) if c4 else ("mu", {})
    if strong:
        V, W, H = make_problem(0)
        c0, nc = na.shard_columns(cols, world, rank)
        Vs, Hs = np.asfortranarray(V[:, c0:c0 + nc]), np.asfortranarray(H[:, c0:c0 + nc])
        total = cols
        full = (V, W, H) if rank == 0 else None
    else:
        Vs, W, Hs = make_problem(rank, rows, cols, feats)
        nc, total = cols, cols * world
        full = (Vs, W, Hs) if rank == 0 else None
    mode = args.shard_mode if args.shard_mode >= 0 else (0 if 4.0 * rows * feats >= 8e6 else 1)
    return dict(c4=c4, strong=strong, rows=rows, cols=cols, feats=feats, alg=alg, alg_kw=alg_kw, V=Vs, W=W, H=Hs, nc=nc, total=total, mode=mode, full=full)


def multi_line(args, pb, world, elapsed, kernel, frob, transport: str, shared_devices: int = 0):
    """The JSON line of an N-GPU run (rank 0)."""
    K, Wm = args.steps, args.warmup
    c4, strong, rows, cols, feats, nc = pb["c4"], pb["strong"], pb["rows"], pb["cols"], pb["feats"], pb["nc"]
    kernel_ms, kernel_launches, pair_overhead_ms = kernel
    bytes_per_launch = (2.0 if c4 else 4.0) * rows * nc    # rank 0's image of its columns of V (bf16 at config 4), one product
    roofline = None
    if kernel_launches > 0:
        avg_s = kernel_ms / 1e3 / kernel_launches
        roofline = {"bound": "hbm", "achieved": bytes_per_launch / avg_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": bytes_per_launch / avg_s / 1e9 / PEAK_HBM_GBS,
                    "traffic": measured_traffic("factor_product_bf16", "nmfgpu_amd/csrc/kernels_bf16.hip")[0] if c4 else measured_traffic("factor_product_x3", "nmfgpu_amd/csrc/kernels_x3.hip")[0],
                    "kernel": "factor product (rank 0's launches)",
                    "avg_launch_us": avg_s * 1e6, "idle_event_pair_us": pair_overhead_ms * 1e3, "launches": kernel_launches, "bytes_per_launch": bytes_per_launch}
    mode_text = ("row-block W step: reduce-scatter of (V H^T)^T by row blocks of W + all-reduce of H H^T, every rank updates its rows, all-reduce of the column norms, all-gather of W"
                 if pb["mode"] == 0 else "replicated W step on the summed (V H^T | H H^T)")
    parallelism = f"column shards x{world}, {transport}: {mode_text}"
    if shared_devices:
        parallelism += f" -- REHEARSAL: {world} ranks share {shared_devices} device(s) (--allow-shared-device), not a scaling measurement"
    out = {"metric": "nsNMF iterations/sec, bf16 operands, dense 50000 x 6250 column shard per GPU, r=256" if c4 else "NMF MU iterations/sec, dense 10kx5k r=64",
           "value": (K if strong else world * K) / elapsed,
           "unit": "iterations/s" if strong else f"shard-iterations/s (one {rows}x{cols} column shard per GPU)",
           "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
           "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "bf16" if c4 else "f32", "data": "synthetic",
           "config": {"workload": ("configs[3]: dense random V 50000 x (6250 per GPU) column shards, r=256, nsNMF theta=0.5, bf16 MFMA operands" if c4 else
                                   "configs[1]: the ONE dense random V 10000x5000, r=64, MU Frobenius, fp32, column-sharded over the GPUs" if strong else
                                   "configs[1]: dense random V 10000x5000 (per GPU), r=64, MU Frobenius, fp32"),
                      "rows": rows, "columns_per_gpu": nc, "total_columns": pb["total"], "features": feats, "error_every": 10,
                      "setup_iterations": SETUP_ITERATIONS // 3 if pb["c4"] else SETUP_ITERATIONS, "parallelism": parallelism},
           "frobenius_last": frob,
           "roofline": whole_iteration(roofline, elapsed / K, floor_bytes=2.0 * bytes_per_launch, basis="rank 0's two passes over its image of V per iteration at the HBM peak")}
    if not args.no_cpu_baseline and pb["full"] is not None:
        phase("cpu baseline")
        Vf, Wf, Hf = pb["full"]
        out["cpu_baseline"] = (cpu_baseline(Vf, Wf, Hf, budget_s=12.0, algorithm="nsnmf", theta=C4["theta"]) if c4 else cpu_baseline(Vf, Wf, Hf))
    return out


def team_worker(args):
    """`bench.py --team-worker`: ONE process, N rank threads, rank g on device g (ranks share devices only in a rehearsal), the in-process transport of
    include/nmfgpu_amd.h.  The timed region: every rank passes a thread barrier with its device idle, runs K iterations of the native sharded loop, waits for its
    stream; elapsed = the longest of the ranks' times."""
    import threading
    import torch
    import nmfgpu_amd as na
    N = args.gpus
    phase("team: counting devices")
    devices = torch.cuda.device_count()
    if devices < 1 or na.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    if devices < N and not args.allow_shared_device:
        print(f"bench.py: --gpus {N} asked for, {devices} HIP device(s) visible", file=sys.stderr, flush=True)
        raise SystemExit(2)
    K, Wm = args.steps, args.warmup
    phase("team: generating the problem")
    problems = [multi_problem(args, N, g) for g in range(N)]
    group = na.LocalGroup(N)
    gate = threading.Barrier(N)
    res = [None] * N
    errors = []

    def rank_thread(g):
        eng = run = comm = None
        try:
            pb = problems[g]
            torch.cuda.set_device(g % devices)
            stream = torch.cuda.Stream()
            if os.environ.get("NMFAMD_BENCH_TEST_HANG") and g == N - 1:
                phase(f"team: rank {g} joining the group (NMFAMD_BENCH_TEST_HANG: never)")
                time.sleep(10 ** 6)
            phase(f"team: rank {g} joining the group")
            comm = na.LocalComm(group, g)                   # blocks until every rank has joined; fails everywhere if two devices cannot map each other
            phase(f"team: rank {g} uploading its shard")
            eng = na.Engine(pb["rows"], pb["nc"], pb["feats"], pb["alg"], dtype=np.float32, stream=stream.cuda_stream, row_blocks=N, **pb["alg_kw"])
            eng.upload(pb["V"])
            eng.set_factors(pb["W"], pb["H"])
            gate.wait()                                     # (every rank holds its engine before anybody enters the run's set-up, which is a collective)
            run = na.ShardedRun(eng, comm, pb["rows"], pb["total"], pb["mode"])
            phase(f"team: rank {g} first iterations (set-up)")
            run.iterate(SETUP_ITERATIONS if not pb["c4"] else SETUP_ITERATIONS // 3, first_iteration=1, error_every=10)
            eng.synchronize()
            eng.set_factors(pb["W"], pb["H"])
            phase(f"team: rank {g} warm-up")
            run.iterate(Wm, first_iteration=1, error_every=10)
            eng.synchronize()
            torch.cuda.synchronize()
            gate.wait()
            phase(f"team: rank {g} timed iterations")
            t0 = time.perf_counter()
            run.iterate(K, first_iteration=Wm + 1, error_every=10)
            eng.synchronize()
            dt = time.perf_counter() - t0
            torch.cuda.synchronize()
            gate.wait()
            kernel = (0.0, 0, 0.0)
            if not args.no_kernel_events:
                # untimed replay behind the timed steps (every rank runs it: the iteration holds a rendezvous); rank 0's product launches are sampled there
                phase(f"team: rank {g} untimed replay (launch samples)")
                replay = replay_iterations(K, pb["c4"])
                if g == 0:
                    eng.kernel_timing(max(1, replay // REPLAY_SAMPLES))
                run.iterate(replay, first_iteration=Wm + K + 1, error_every=10)
                eng.synchronize()
                if g == 0:
                    kernel = eng.kernel_timing_read2()
                gate.wait()
            res[g] = (dt, kernel, run.frobenius if g == 0 else 0.0)
        except BaseException as e:                          # noqa: BLE001 -- a failed rank releases its peers (collectives, the gate) and the process exits non-zero
            errors.append((g, e))
            group.abort()
            gate.abort()
        finally:
            for obj in (run, eng, comm):
                if obj is not None:
                    try:
                        obj.close()
                    except Exception:                       # noqa: BLE001
                        pass

    threads = [threading.Thread(target=rank_thread, args=(g,), daemon=True) for g in range(N)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    real = [(g, e) for g, e in errors if not isinstance(e, threading.BrokenBarrierError)]
    if errors:
        for g, e in (real or errors):
            print(f"bench.py: team rank {g} failed in phase '{_PHASE['name']}': {e!r}", file=sys.stderr, flush=True)
        raise SystemExit(4)
    elapsed = max(r[0] for r in res)
    selftest = group.selftest_report() or "peer-transport self-test: not run"
    out = multi_line(args, problems[0], N, elapsed, res[0][1], res[0][2],
                     "in-library team (one process, one rank thread per GPU; in-process transport: the ranks' kernels read each other's buffers in place, peer access over xGMI; "
                     + selftest + ")",
                     shared_devices=devices if devices < N else 0)
    print(json.dumps(out), flush=True)


def rccl_ranks(args):
    """One process per GPU under torch.distributed.run: rank 0 first tries the team (--transport auto) as a child process while the other ranks wait; otherwise, or
    when that fails, every rank runs the native sharded loop with RCCL through its C API.  torch.distributed (gloo) carries the unique id, barriers and timings."""
    import datetime
    import torch
    import torch.distributed as dist
    world = int(os.environ["WORLD_SIZE"]); rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    start_watchdog(f"rank {rank}")
    phase("control group (gloo)")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")          # one node: the box's hostname may not resolve
    dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=deadline_seconds()))
    devices = torch.cuda.device_count()                        # (does not initialise the GPU)
    if args.transport in ("auto", "team") or devices < world:
        phase("rank 0 runs the team as a child process")
        box = [None]
        if rank == 0:
            ok, line, rc = launch_team(args, max(5.0, 0.6 * deadline_seconds()))
            box[0] = (ok, line, rc)
        dist.broadcast_object_list(box, src=0)
        ok, line, rc = box[0]
        if ok:
            if rank == 0:
                print(line, flush=True)
            dist.barrier()
            dist.destroy_process_group()
            return
        if args.transport == "team" or devices < world:
            dist.destroy_process_group()
            raise SystemExit(rc or 1)
        if rank == 0:
            print("bench.py: the team could not be set up; one process per GPU over RCCL", file=sys.stderr, flush=True)
    import nmfgpu_amd as na
    phase("RCCL: device and library")
    if not torch.cuda.is_available() or na.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank % devices)
    pb = multi_problem(args, world, rank)
    comm = eng = run = None
    failure = None if na.RcclComm.available() else RuntimeError("librccl.so could not be loaded")

    def agree(failure):
        flag = torch.tensor([0 if failure is None else 1], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return int(flag.item()) != 0

    phase("RCCL: unique id")
    uid = [None]
    if rank == 0 and failure is None:
        try:
            uid[0] = na.RcclComm.unique_id()
        except Exception as e:                    # noqa: BLE001
            failure = e
    dist.broadcast_object_list(uid, src=0)
    if uid[0] is None and failure is None:
        failure = RuntimeError("rank 0 could not create a communicator id")
    if not agree(failure):
        try:
            phase("RCCL: communicator (blocks until every rank has joined)")
            comm = na.RcclComm(uid[0], world, rank)
            phase("RCCL: uploading the shard")
            eng = na.Engine(pb["rows"], pb["nc"], pb["feats"], pb["alg"], dtype=np.float32, stream=engine_stream(torch), row_blocks=world, **pb["alg_kw"])
            eng.upload(pb["V"])
            eng.set_factors(pb["W"], pb["H"])
        except Exception as e:                    # noqa: BLE001
            failure = e
        # every rank must hold its engine before anybody enters the sharded run's set-up (a collective on the engine's stream)
        if not agree(failure):
            try:
                phase("RCCL: sharded run set-up (first collective)")
                run = na.ShardedRun(eng, comm, pb["rows"], pb["total"], pb["mode"])
            except Exception as e:                # noqa: BLE001
                failure = e
    if agree(failure):
        print(f"bench.py: RCCL set-up failed on rank {rank}: {failure!r}" if failure is not None else f"bench.py: RCCL set-up failed on another rank (this is rank {rank})",
              file=sys.stderr, flush=True)
        for obj in (run, eng, comm):
            if obj is not None:
                obj.close()
        dist.destroy_process_group()
        raise SystemExit(5)
    K, Wm = args.steps, args.warmup

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    # set-up, not steps: the first collectives of a communicator and the first launches of every kernel
    phase("RCCL: first iterations (set-up)")
    run.iterate(SETUP_ITERATIONS if not pb["c4"] else SETUP_ITERATIONS // 3, first_iteration=1, error_every=10)
    eng.synchronize()
    eng.set_factors(pb["W"], pb["H"])
    phase("RCCL: warm-up")
    run.iterate(Wm, first_iteration=1, error_every=10)
    eng.synchronize()
    barrier()
    phase("RCCL: timed iterations")
    t0 = time.perf_counter()
    run.iterate(K, first_iteration=Wm + 1, error_every=10)
    eng.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    kernel = (0.0, 0, 0.0)
    if not args.no_kernel_events:
        # untimed replay behind the timed steps (a collective: every rank runs it); this rank's product launches are sampled there
        phase("RCCL: untimed replay (launch samples)")
        replay = replay_iterations(K, pb["c4"])
        eng.kernel_timing(max(1, replay // REPLAY_SAMPLES))
        run.iterate(replay, first_iteration=Wm + K + 1, error_every=10)
        eng.synchronize()
        kernel = eng.kernel_timing_read2()
        barrier()
    frob = run.frobenius
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        print(json.dumps(multi_line(args, pb, world, elapsed, kernel, frob, "one process per GPU, native loop with RCCL through its C API")), flush=True)
    run.close(); eng.close(); comm.close()
    phase("closing")
    dist.barrier()
    dist.destroy_process_group()



PEAK_FP64_MFMA_TFLOPS = 78.6    # /opt/skills/guides (kernels_f64.hip header): the fp64 matrix peak of gfx950 equals its vector peak, 78.6 TFLOP/s
EXAMPLE = {"rows": 4096, "columns": 165, "features": 158, "theta": 0.5}      # /root/reference example/main.cpp:31-33,119-126


def main_f64(args):
    """The double-precision instantiation -- what the reference's own callers run (example/main.cpp: NmfDescription<double>; the R binding is double by nature).
    c2-f64: config 2's shape, MU, r = 64.  example: the reference's example program's problem (4096 x 165, r = 158, nsNMF theta = 0.5).  Products on
    v_mfma_f64_16x16x4_f64 (kernels_f64.hip); roofline: the factor product against the fp64 MFMA peak.  Same timing contract as the default workload."""
    import torch
    import nmfgpu_amd as na
    if not torch.cuda.is_available() or na.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    torch.cuda.set_device(0)
    ex = args.workload == "example"
    if ex:
        m, n, r, alg, kw = EXAMPLE["rows"], EXAMPLE["columns"], EXAMPLE["features"], "nsnmf", dict(theta=EXAMPLE["theta"])
        rs = np.random.RandomState(1)
        # the example's data: multiples of 1/255 in [0, 254/255] (rand() % 255 / 255.0); start values U(0, 1] as everywhere here (the example draws AllRandomValues)
        V = np.asfortranarray(rs.randint(0, 255, size=(m, n)).astype(np.float64) / 255.0)
        W = np.asfortranarray(1.0 - rs.random_sample((m, r)))
        H = np.asfortranarray(1.0 - rs.random_sample((r, n)))
        setup = 2000
    else:
        m, n, r, alg, kw = M, N_COLS, R, "mu", {}
        V32, W32, H32 = make_problem(0)
        V, W, H = (np.asfortranarray(x.astype(np.float64)) for x in (V32, W32, H32))
        del V32
        setup = SETUP_ITERATIONS // 2
    K, Wm = args.steps, args.warmup
    eng = na.Engine(m, n, r, alg, dtype=np.float64, stream=engine_stream(torch), **kw)
    eng.upload(V)
    eng.set_factors(W, H)
    eng.iterate(setup, first_iteration=1, error_every=10)       # set-up (first launches, the device's ramp from idle: see main()), then back to W0, H0
    eng.synchronize()
    eng.set_factors(W, H)
    eng.iterate(Wm, first_iteration=1, error_every=10)
    eng.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.iterate(K, first_iteration=Wm + 1, error_every=10)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms, kernel_launches, pair_overhead_ms = 0.0, 0, 0.0
    if not args.no_kernel_events:
        replay = replay_iterations(K, False)
        eng.kernel_timing(max(1, replay // REPLAY_SAMPLES))
        eng.iterate(replay, first_iteration=Wm + K + 1, error_every=10)
        eng.synchronize()
        kernel_ms, kernel_launches, pair_overhead_ms = eng.kernel_timing_read2()
        eng.kernel_timing(0)
    frob = eng.frobenius
    flops_per_launch = 2.0 * m * n * r
    bytes_per_launch = 8.0 * m * n
    roofline = None
    if kernel_launches > 0:
        avg_s = kernel_ms / 1e3 / kernel_launches
        ach = flops_per_launch / avg_s / 1e12
        # (counter traffic of the product launch WITH its Gram passengers, measured at config 2's shape: tools/pmc_traffic.py, sha-stamped; null for the example's shape)
        f64_traffic, f64_traffic_source = measured_traffic("factor_product_f64", "nmfgpu_amd/csrc/kernels_f64.hip") if not ex else (None, "not measured at this shape")
        roofline = {"bound": "mfma", "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP64_MFMA_TFLOPS, "traffic": f64_traffic,
                    "traffic_source": f64_traffic_source,
                    "kernel": "k_factor_product_f64", "launch_also_carries": "the Gram matrix of the operand as passenger workgroups (kernels_f64.hip, gram_ride_f64)",
                    "avg_launch_us": avg_s * 1e6, "idle_event_pair_us": pair_overhead_ms * 1e3, "launches": kernel_launches,
                    "flops_per_launch": flops_per_launch, "bytes_per_launch": bytes_per_launch,
                    "timed_by": "an event recorded before and one after the launch (over-reports by about idle_event_pair_us), " + TIMED_BY_REPLAY,
                    "hbm_side": {"achieved": bytes_per_launch / avg_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": bytes_per_launch / avg_s / 1e9 / PEAK_HBM_GBS}}
    iter_flops = 4.0 * m * n * r + (6.0 if ex else 4.0) * r * r * (m + n)
    out = {"metric": ("nsNMF iterations/sec, double precision, the reference example's 4096x165 r=158" if ex else "NMF MU iterations/sec, double precision, dense 10kx5k r=64"),
           "value": K / elapsed, "unit": "iterations/s", "n_gpus": 1, "steps": K, "warmup": Wm, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": ("the reference's example program (example/main.cpp): dense V 4096x165 of multiples of 1/255, r=158, nsNMF theta=0.5, double" if ex else
                                   "configs[1]'s shape in double precision: dense random V 10000x5000, r=64, MU Frobenius, fp64"),
                      "rows": m, "columns": n, "features": r, "error_every": 10, "setup_iterations": setup, "parallelism": "single GPU",
                      "arithmetic": "fp64 throughout; products on v_mfma_f64_16x16x4_f64",
                      # (round 6) 4: product + Gram passengers / update / product + Gram passengers / update (Engine::iterate_fused64); 0: the generic launch sequence
                      "launches_per_iteration": eng.geometry()["fused_launches"] or None,
                      "gram_passenger_k_slices": [eng.geometry()["gram_ride_slices_h"], eng.geometry()["gram_ride_slices_w"]]},
           "frobenius_last": frob, "iter_flops": iter_flops, "achieved_tflops_whole_iteration": iter_flops * (K / elapsed) / 1e12,
           "roofline": whole_iteration(roofline, elapsed / K, floor_flops=2.0 * flops_per_launch, basis="the two products against V at the fp64 MFMA peak")}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(V, W, H, budget_s=15.0, algorithm=alg, **kw)
    print(json.dumps(out), flush=True)

def main_c3(args):
    """BASELINE configs[2] on one GPU: CSR V 100000 x 20000 at 1 % (2e7 stored entries), r = 128, multiplicative update on
    the generalised KL divergence; V stays sparse in HBM (CSR + CSC images), one step = one iteration = two fused half-steps
    (quotient V ./ (W H) and numerator in one gather pass each).  Same timing contract as the default.  roofline: that kernel."""
    import torch
    import nmfgpu_amd as na
    if not torch.cuda.is_available() or na.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    torch.cuda.set_device(0)
    m, n, r = C3["rows"], C3["columns"], C3["features"]
    val, ptr, idx, W, H = make_sparse_problem()
    nnz = int(len(val))
    K, Wm = args.steps, args.warmup
    eng = na.Engine(m, n, r, "mu", dtype=np.float32, stream=engine_stream(torch), divergence="kl")
    eng.upload_sparse(1, val, ptr, idx, 0)
    eng.set_factors(W, H)
    eng.iterate(SETUP_ITERATIONS // 10, first_iteration=1, error_every=10)      # set-up (first launches, the device's ramp from idle: see main()), then back to W0, H0
    eng.synchronize()
    eng.set_factors(W, H)
    eng.iterate(Wm, first_iteration=1, error_every=10)
    eng.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.iterate(K, first_iteration=Wm + 1, error_every=10)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms, kernel_launches, pair_overhead_ms = 0.0, 0, 0.0
    if not args.no_kernel_events:
        replay = max(10, min(K, 30))       # untimed replay behind the timed steps: the launch samples of `roofline`
        eng.kernel_timing(max(1, replay // 5))
        eng.iterate(replay, first_iteration=Wm + K + 1, error_every=10)
        eng.synchronize()
        kernel_ms, kernel_launches, pair_overhead_ms = eng.kernel_timing_read2()
        eng.kernel_timing(0)
    rp = eng.geometry()["padded_rank"]
    # per launch of the fused half-step kernel (two per iteration; SURVEY 8d's accounting): value + index of every stored entry
    # in, the row pointers, one pass over both factors, the numerator panel out (average of the H and the W half-step)
    hbm_bytes = 8.0 * nnz + 2.0 * (m + n + 2) + 4.0 * rp * (m + n) + 2.0 * rp * (m + n)
    gather_bytes = 4.0 * rp * nnz                      # one 512-byte factor row per stored entry, served from cache
    roofline = None
    if kernel_launches > 0:
        avg_s = kernel_ms / 1e3 / kernel_launches
        c3_traffic, c3_traffic_source = measured_traffic("kl_fused", "nmfgpu_amd/csrc/kernels_sparse.hip")      # what leaves L2: the row gathers that miss it, mostly served by the memory-side cache
        roofline = {"bound": "hbm", "achieved": hbm_bytes / avg_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": hbm_bytes / avg_s / 1e9 / PEAK_HBM_GBS, "traffic": c3_traffic, "traffic_source": c3_traffic_source, "kernel": "k_kl_fused",
                    "avg_launch_us": avg_s * 1e6, "idle_event_pair_us": pair_overhead_ms * 1e3, "launches": kernel_launches, "bytes_per_launch": hbm_bytes,
                    "note": ("the binding resource is the cache-level gather of factor rows, not HBM (SURVEY 8d): see `gather`; `traffic` = bytes that leave L2 "
                             "(FETCH_SIZE x 2 + WRITE_SIZE): the row gathers that miss L2, served by the memory-side cache which holds both factors"),
                    "gather": {"achieved": gather_bytes / avg_s / 1e9, "peak": PEAK_L2_GATHER_GBS, "unit": "GB/s",
                               "frac": gather_bytes / avg_s / 1e9 / PEAK_L2_GATHER_GBS, "bytes_per_launch": gather_bytes}}
    out = {"metric": "NMF KL-divergence MU iterations/sec, sparse CSR 100kx20k 1% r=128",
           "value": K / elapsed, "unit": "iterations/s", "n_gpus": 1, "steps": K, "warmup": Wm, "ms_per_step": elapsed / K * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[2]: sparse CSR V 100000x20000 at 1% (Poisson(200) entries per row, values 1..5), r=128, MU on the KL divergence, fp32",
                      "rows": m, "columns": n, "features": r, "stored_entries": nnz, "error_every": 10, "setup_iterations": SETUP_ITERATIONS // 10, "parallelism": "single GPU"},
           "frobenius_last": eng.frobenius, "kl_divergence_last": eng.kl_divergence,
           "iter_flops": 8.0 * nnz * r, "achieved_tflops_whole_iteration": 8.0 * nnz * r * (K / elapsed) / 1e12,
           "gather_gbs_whole_iteration": 2.0 * gather_bytes * (K / elapsed) / 1e9, "roofline": roofline}
    if roofline is not None:
        roofline["whole_iteration_frac"] = (2.0 * gather_bytes / (PEAK_L2_GATHER_GBS * 1e9)) / (elapsed / K)
        roofline["whole_iteration_basis"] = "two gather passes over the stored entries (one 512-byte factor row each) at the L2 row-gather rate"
    if not args.no_cpu_baseline:
        from oracle import oracle
        Wc, Hc = W.copy(order="F"), H.copy(order="F")
        t0 = time.perf_counter()
        oracle.run_kl_csr(m, n, val, ptr, idx, Wc, Hc, 1)
        first = time.perf_counter() - t0
        iters = int(max(1, min(10, 20.0 // max(first, 1e-3))))
        t0 = time.perf_counter()
        oracle.run_kl_csr(m, n, val, ptr, idx, Wc, Hc, iters)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": iters / dt, "unit": "iterations/s", "cores": oracle.num_threads(), **host_cpu_info(), "kind": "port",
                               "sample": f"{iters} KL-MU iterations of the full problem over the stored entries (oracle_kl_run_csr, C + OpenMP, fp32)"}
    print(json.dumps(out), flush=True)


def main_c4(args):
    """BASELINE configs[3] per GPU: 50000 x 6250 column shard of a 50000 x (6250 N) matrix, nsNMF theta = 0.5, r = 256,
    bf16 MFMA operands, W replicated, all-reduce of 51.2 MB + 256 KB per iteration.  Same timing contract as the default."""
    import torch
    import nmfgpu_amd as na
    from nmfgpu_amd.distributed import EngineShard, ShardedMU
    world, rank = 1, 0
    if not torch.cuda.is_available() or na.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    torch.cuda.set_device(0)
    m, n, r, theta = C4["rows"], C4["columns_per_gpu"], C4["features"], C4["theta"]
    V, W, H = make_problem(rank, m, n, r)
    K, Wm = args.steps, args.warmup
    shard = EngineShard(V, W, H, algorithm="nsnmf", theta=theta, precision="bf16")
    drv = ShardedMU(shard, total_columns=n * world, rows=m)
    drv.run(SETUP_ITERATIONS // 3, first_iteration=1, error_every=10)      # set-up (one-time costs, the device's ramp from idle: see main()), then back to W0, H0
    shard.synchronize()
    shard.engine.set_factors(W, H)

    def barrier():
        torch.cuda.synchronize()

    drv.run(Wm, first_iteration=1, error_every=10)
    shard.synchronize()
    barrier()
    t0 = time.perf_counter()
    drv.run(K, first_iteration=Wm + 1, error_every=10)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, kernel_launches, pair_overhead_ms = 0.0, 0, 0.0
    if not args.no_kernel_events:
        replay = replay_iterations(K, True)       # untimed replay behind the timed steps: the launch samples of `roofline`
        shard.engine.kernel_timing(max(1, replay // REPLAY_SAMPLES))
        drv.run(replay, first_iteration=Wm + K + 1, error_every=10)
        shard.synchronize()
        kernel_ms, kernel_launches, pair_overhead_ms = shard.engine.kernel_timing_read2()
        shard.engine.kernel_timing(0)
    if rank == 0:
        bytes_per_launch = 2.0 * m * n                       # one pass over the bf16 image of the shard
        roofline = None
        if kernel_launches > 0:
            avg_s = kernel_ms / 1e3 / kernel_launches
            roofline = {"bound": "hbm", "achieved": bytes_per_launch / avg_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": bytes_per_launch / avg_s / 1e9 / PEAK_HBM_GBS, "traffic": measured_traffic("factor_product_bf16", "nmfgpu_amd/csrc/kernels_bf16.hip")[0], "kernel": "k_factor_product_bf16",
                        "avg_launch_us": avg_s * 1e6, "idle_event_pair_us": pair_overhead_ms * 1e3, "launches": kernel_launches, "bytes_per_launch": bytes_per_launch,
                        "launch_also_carries": "32 passenger workgroups: the 256 x 256 Gram matrix of the launch's factor operand, its diagonal and split image (tri_gram_tile.h): "
                                               "a product launch is 10-15 us longer than alone, the iteration 14 us shorter; bytes_per_launch counts the product's stream of V only"}
        iter_flops = 4.0 * m * n * r + 6.0 * r * r * (m + n)
        print(json.dumps({
            "metric": "nsNMF iterations/sec, bf16 operands, dense 50000 x 6250 column shard per GPU, r=256",
            "value": world * K / elapsed, "unit": "shard-iterations/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "configs[3] per GPU: dense random V 50000x6250 column shard, r=256, nsNMF theta=0.5, bf16 MFMA operands",
                       "rows": m, "columns_per_gpu": n, "features": r, "error_every": 10, "setup_iterations": SETUP_ITERATIONS // 3,
                       "parallelism": f"column shards x{world}, W replicated, all-reduce of (V (SH)^T | (SH)(SH)^T) per iteration"},
            "frobenius_last": drv.frobenius, "iter_flops": iter_flops,
            "achieved_tflops_whole_iteration": iter_flops * (K / elapsed) / 1e12,
            "roofline": whole_iteration(roofline, elapsed / K, floor_bytes=2.0 * bytes_per_launch, basis="two passes over the bf16 image of the shard per iteration (2 x 625 MB) at the HBM peak"),
            **({} if args.no_cpu_baseline else {"cpu_baseline": cpu_baseline(V, W, H, budget_s=12.0, algorithm="nsnmf", theta=theta)})}), flush=True)


if __name__ == "__main__":
    main()
