"""What an error iteration costs the resident engine at config 2: iterations with error_every = 10 / 0 / 1 (round 5, end: 87.7 / 87.1 / 101 us per iteration on one box --
5.5 us per error iteration at every tenth, 13.5 when every iteration is one)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, bench, nmfgpu_amd as na
V, W, H = bench.make_problem(0)
e = na.Engine(V.shape[0], V.shape[1], W.shape[1], "mu"); e.upload(V); e.set_factors(W, H)
e.iterate(400, first_iteration=1, error_every=10); e.synchronize()
for rep in range(3):
    for ee in (10, 0, 1):
        t0 = time.perf_counter(); e.iterate(400, first_iteration=1, error_every=ee); e.synchronize()
        print(f"error_every={ee}: {(time.perf_counter()-t0)/400*1e6:.2f} us/iteration")
