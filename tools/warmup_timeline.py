"""Iteration time of config 2 in chunks of 25 from the first iteration of a process, and again after 2 s of idling: the ramp bench.py's set-up iterations cover."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import nmfgpu_amd as na
V, W, H = bench.make_problem(0)
e = na.Engine(bench.M, bench.N_COLS, bench.R, "mu")
e.upload(V); e.set_factors(W, H)
e.synchronize()
out = []
it = 1
t_start = time.perf_counter()
for c in range(40):
    t0 = time.perf_counter()
    e.iterate(25, first_iteration=it, error_every=10); e.synchronize()
    out.append((time.perf_counter() - t0) / 25 * 1e6); it += 25
print("us/iteration per chunk of 25:", " ".join(f"{x:.1f}" for x in out))
time.sleep(2.0)
out = []
for c in range(12):
    t0 = time.perf_counter()
    e.iterate(25, first_iteration=it, error_every=10); e.synchronize()
    out.append((time.perf_counter() - t0) / 25 * 1e6); it += 25
print("after 2 s idle:", " ".join(f"{x:.1f}" for x in out))
