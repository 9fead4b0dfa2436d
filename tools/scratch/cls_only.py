"""Runs the classifier alone on 1024 random crops (profiling target)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import retto_amd
s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
x = np.random.default_rng(1).uniform(-1, 1, (1024, 3, 48, 192)).astype(np.float32)
for _ in range(3):
    y = s.worker.cls(x)
print(y[:2])
