"""bench.py's one-line JSON contract, on the GPU: a short default run must print exactly one JSON object with the metric,
the whole-job value, the roofline object of the dominant kernel (live HIP-event timing) and the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "10"], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"].startswith("NMF MU iterations/sec") and d["unit"] == "iterations/s"
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 10 and d["higher_is_better"] is True
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["scaling"] == "strong"
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-6) and d["value"] > 1000
    cfg = d["config"]
    assert cfg["rows"] == 10000 and cfg["columns"] == 5000 and cfg["features"] == 64 and "model" not in cfg and "per GPU" not in cfg["workload"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9) and 0.05 < r["frac"] < 1.2
    assert r["launches"] > 0 and r["avg_launch_us"] > 10
    assert "untimed replay" in r["timed_by"]             # (round 5: the timed region carries no events; the launch samples come from a replay right behind it)
    if r["bound"] == "hbm":
        assert r["achieved"] == pytest.approx(r["bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9, rel=1e-6)
    if r["kernel"] == "k_factor_product_x3":
        # the two forms of the kernel are stamped apart (W^T V reads the one image along the reduction index, V H^T along its output index), and the same
        # launches are priced against the bf16 matrix pipe as well: neither resource is saturated, the line says so instead of implying an HBM-bound kernel
        f = r["forms"]
        assert set(f) == {"wt_v", "v_ht"} and all(v["launches"] > 0 and v["frac"] == pytest.approx(v["achieved"] / r["peak"], rel=1e-9) for v in f.values())
        assert f["wt_v"]["launches"] + f["v_ht"]["launches"] == r["launches"] and "memory-side cache" in r["streams_from"]
        m2 = d["roofline_mfma"]
        assert m2["bound"] == "mfma" and m2["unit"] == "TFLOP/s" and m2["peak"] == 2500.0 and m2["frac"] == pytest.approx(m2["achieved"] / m2["peak"], rel=1e-9)
        assert m2["achieved"] == pytest.approx(6.0 * r["flops_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12, rel=1e-6) and 0.1 < m2["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and isinstance(c["sample"], str)
    # round 6: the line says what the CPU legs ran on (probed: affinity mask, cgroup quota, /proc/cpuinfo) and uses every available core unless --cpu-threads caps it
    assert c["affinity_cores"] >= c["cores_available"] >= 1 and c["cores"] == c["cores_available"] and isinstance(c["cpu_model"], str) and c["cpu_model"]
    assert c["cgroup_quota_cores"] is None or c["cgroup_quota_cores"] > 0
    # ... and carries BOTH warm states: `value` behind the harness's set-up phase (config.setup_iterations), value_driver_warmup_only = the same K steps right
    # behind the driver's --warmup steps and nothing else (a cold device runs slower: the ramp from idle takes ~125 iterations, profiles/r04_warmup_timeline.txt)
    assert d["config"]["setup_iterations"] > 0
    assert d["value_driver_warmup_only"] == pytest.approx(1e3 / d["ms_per_step_driver_warmup_only"], rel=1e-6)
    assert 0.5 * d["value"] < d["value_driver_warmup_only"] < 1.1 * d["value"]
    # the iteration converged to the same error as every other path on this input (oracle: 2022.63 after 220 iterations;
    # here after 50)
    assert 2000 < d["frobenius_last"] < 2100



def test_gpus_n_without_torchrun_refuses_or_rehearses():
    """`python bench.py --gpus 2` outside torchrun on a box with ONE device: no line at all and a non-zero exit (never an n_gpus = 1
    line under a --gpus 2 command); with --allow-shared-device it runs the in-library team itself (two rank threads share the device) and the
    line is the north_star's series -- the ONE 10000 x 5000 problem, "scaling": "strong", n_gpus = 2 -- and says that it is a rehearsal."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-device box")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"]
    out = subprocess.run(base, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "--gpus 2" in out.stderr and "1 HIP device" in out.stderr
    out = subprocess.run(base + ["--allow-shared-device"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and "REHEARSAL" in d["config"]["parallelism"] and "team" in d["config"]["parallelism"]
    assert d["scaling"] == "strong" and d["unit"] == "iterations/s" and d["config"]["total_columns"] == 5000 and d["config"]["columns_per_gpu"] == 2500
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-6)
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"            # rank 0 reports it for every N
    assert 2000 < d["frobenius_last"] < 2200                                                # the same problem as the N = 1 line (7 iterations in)
    # --scaling weak is the explicit other series: one 10000 x 5000 shard per rank
    out = subprocess.run(base + ["--allow-shared-device", "--scaling", "weak", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "weak" and d["config"]["total_columns"] == 10000 and d["value"] == pytest.approx(2 * 1e3 / d["ms_per_step"], rel=1e-6)


def test_a_hung_rank_ends_the_bench_inside_the_deadline_with_the_phase():
    """First contact with a node nobody has run on: a rank that never joins must not hang the bench until the driver's timeout.  NMFAMD_BENCH_TEST_HANG=1 makes
    the last rank thread of the team sleep instead of joining; the launcher kills the child process group at the deadline, names the phase, exits non-zero."""
    import time
    env = dict(os.environ, NMFAMD_BENCH_TEST_HANG="1", NMFAMD_BENCH_DEADLINE="45")
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--allow-shared-device", "--no-cpu-baseline",
                          "--transport", "team"], cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    took = time.monotonic() - t0
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "deadline" in out.stderr and "phase" in out.stderr and "joining the group" in out.stderr, out.stderr[-2000:]
    assert took < 120
