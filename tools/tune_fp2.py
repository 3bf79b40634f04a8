import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tune_fp import run
for X, Y in [(2560, 5000), (1280, 10000), (10000, 5000), (20000, 5000), (10000, 10000)]:
    run(X, Y, reps=30, stamps=True)
