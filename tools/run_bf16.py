import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmfgpu_amd as na
V = np.asfortranarray(np.random.RandomState(1).random_sample((5000, 10000)).astype(np.float32).T)
W = np.asfortranarray((1.0 - np.random.RandomState(2).random_sample((64, 10000))).astype(np.float32).T)
H = np.asfortranarray((1.0 - np.random.RandomState(3).random_sample((5000, 64))).astype(np.float32).T)
eng = na.Engine(10000, 5000, 64, "mu", precision="bf16")
eng.upload(V); eng.set_factors(W, H)
eng.iterate(100, first_iteration=1); eng.synchronize()
print(eng.frobenius)
