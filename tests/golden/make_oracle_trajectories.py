"""Generates tests/golden/oracle_trajectories.json: error trajectories and factor checksums of the ORACLE (our CPU
restatement) on small seeded problems, in double precision, for every algorithm and the KL extension.

These vectors do not come from the reference (it cannot run here): they freeze the oracle's own
behaviour, so that a change to the oracle (blocking, threading, summation order) that moves its results by more than
rounding is caught by tests/test_oracle.py.  Re-run in the build container:
    python tests/golden/make_oracle_trajectories.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

CASES = [("mu", {}), ("nsnmf", dict(theta=0.4)), ("gdcls", dict(lam=0.05)), ("als", {}), ("acls", dict(lambda_w=0.1, lambda_h=0.2)),
         ("ahcls", dict(lambda_w=0.1, lambda_h=0.2, alpha_w=0.3, alpha_h=0.4))]


def problem(seed, m=61, n=47, r=5):
    rng = np.random.default_rng(seed)
    return (np.asfortranarray(rng.random((m, n))), np.asfortranarray(1.0 - rng.random((m, r))), np.asfortranarray(1.0 - rng.random((r, n))))


def main():
    out = {"generator": "tests/golden/make_oracle_trajectories.py", "dtype": "float64", "shape": [61, 47, 5], "iterations": 40, "cases": []}
    for k, (alg, kw) in enumerate(CASES):
        V, W, H = problem(100 + k)
        res = oracle.run(alg, V, W, H, 40, **kw)
        out["cases"].append({"algorithm": alg, "parameters": kw, "seed": 100 + k, "history": [[float(f), float(r)] for f, r in res["history"]],
                             "w_sum": float(W.sum()), "h_sum": float(H.sum()), "w_sq": float((W * W).sum()), "h_sq": float((H * H).sum())})
    V, W, H = problem(200)
    V = np.asfortranarray(np.floor(V * 6.0) * (V > 0.6))          # sparse, rating-like
    res = oracle.run_kl(V, W, H, 40)
    out["kl"] = {"seed": 200, "frobenius": res["frobenius"], "kl": res["kl"], "w_sum": float(W.sum()), "h_sum": float(H.sum())}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_trajectories.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
