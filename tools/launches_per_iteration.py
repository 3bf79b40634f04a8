"""usage: python tools/launches_per_iteration.py M N R ALG DTYPE [theta]      (on the GPU box; ALG mu | nsnmf | gdcls | als | acls | ahcls, DTYPE f32 | f64)
Kernel launches and kernel time per steady-state iteration of the resident engine at any shape: two child runs under `rocprofv3 --kernel-trace` (100 and 300
iterations, error terms every 10th), the difference divided by 200 -- set-up, upload and the first launches cancel; a third child without the profiler gives the wall time per iteration.  Prints the per-kernel table of the difference."""
import collections
import csv
import glob
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(m, n, r, alg, dtype, theta, iters):
    sys.path.insert(0, ROOT)
    import numpy as np
    import nmfgpu_amd as na
    dt = np.float32 if dtype == "f32" else np.float64
    rng = np.random.default_rng(1)
    V = np.asfortranarray(rng.random((m, n)).astype(dt))
    W = np.asfortranarray((1.0 - rng.random((m, r))).astype(dt))
    H = np.asfortranarray((1.0 - rng.random((r, n))).astype(dt))
    kw = {"nsnmf": dict(theta=theta), "gdcls": dict(lam=0.01), "acls": dict(lambda_w=0.01, lambda_h=0.01),
          "ahcls": dict(lambda_w=0.01, lambda_h=0.01, alpha_w=0.01, alpha_h=0.01)}.get(alg, {})
    eng = na.Engine(m, n, r, alg, dtype=dt, **kw)
    eng.upload(V); eng.set_factors(W, H)
    eng.iterate(iters, first_iteration=1, error_every=10)
    eng.synchronize()
    t0 = time.perf_counter()
    eng.iterate(200, first_iteration=iters + 1, error_every=10)
    eng.synchronize()
    print(f"WALL {(time.perf_counter() - t0) / 200 * 1e6:.2f}", flush=True)


def trace(args, iters):
    d = tempfile.mkdtemp(prefix="lpi_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", "python3", os.path.abspath(__file__), "--child", str(iters), *args]
    out = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        raise SystemExit(out.stderr[-2000:])
    wall = [float(l.split()[1]) for l in out.stdout.splitlines() if l.startswith("WALL")]
    per = collections.defaultdict(lambda: [0, 0])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = per[row["Kernel_Name"].split("(")[0]]
            k[0] += 1; k[1] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    return per, (wall[0] if wall else float("nan"))


def main():
    if sys.argv[1] == "--child":
        iters = int(sys.argv[2]); m, n, r = (int(x) for x in sys.argv[3:6]); alg, dtype = sys.argv[6], sys.argv[7]
        return child(m, n, r, alg, dtype, float(sys.argv[8]) if len(sys.argv) > 8 else 0.5, iters)
    args = sys.argv[1:]
    a, _ = trace(args, 100)
    b, _ = trace(args, 300)
    out = subprocess.run(["python3", os.path.abspath(__file__), "--child", "300", *args], capture_output=True, text=True, timeout=600)   # no profiler attached
    wall = next((float(l.split()[1]) for l in out.stdout.splitlines() if l.startswith("WALL")), float("nan"))
    rows = []
    for name in b:
        dc = (b[name][0] - a.get(name, [0, 0])[0]) / 200.0
        dns = (b[name][1] - a.get(name, [0, 0])[1]) / 200.0
        if abs(dc) > 1e-9:
            rows.append((dns, dc, name))
    rows.sort(reverse=True)
    print(f"{' '.join(args)}: {sum(r[1] for r in rows):.2f} launches and {sum(r[0] for r in rows) / 1e3:.1f} us of kernel time per iteration; unprofiled wall {wall:.1f} us per iteration")
    for dns, dc, name in rows:
        print(f"  {name[:90]:92s} {dc:6.2f} x {dns / max(dc, 1e-9) / 1e3:8.2f} us")


if __name__ == "__main__":
    main()
