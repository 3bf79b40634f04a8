"""Config 2 on a dedicated stream, eager launches (and, with a library built with the experiment, a captured iteration replayed as a HIP graph:
97 us per iteration against 92 eager -- round 3, not kept).  usage: python tools/graph_experiment.py   (needs a GPU)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import torch, nmfgpu_amd as na
rng = np.random.default_rng(1)
m, n, r = 10000, 5000, 64
V = np.asfortranarray(rng.random((m, n)).astype(np.float32)); W = np.asfortranarray((1 - rng.random((m, r))).astype(np.float32)); H = np.asfortranarray((1 - rng.random((r, n))).astype(np.float32))
st = torch.cuda.Stream()
eng = na.Engine(m, n, r, "mu", stream=st.cuda_stream)
eng.upload(V); eng.set_factors(W, H)
eng.iterate(50, first_iteration=1, error_every=10); eng.synchronize()
for rep in range(3):
    t = time.perf_counter(); eng.iterate(400, first_iteration=51, error_every=10); eng.synchronize(); dt = time.perf_counter() - t
    print(os.environ.get("NMFAMD_GRAPH_EXPERIMENT", "eager"), "%.2f us/iter" % (dt / 400 * 1e6), eng.frobenius, flush=True)
