"""Founder draw of reset(): time bh_founders_device for the configs[1] shape (16 envs, 256 of 370)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import HipEngine
eng = HipEngine(np.full(64, 0.01, np.float32), np.ones((64, 1), np.float32))
key = np.array([0, 3], np.uint32)
for N, n, E in ((370, 256, 16), (2000, 2000, 4), (8192, 819, 8)):
    for _ in range(5):
        eng.founders(key, E, N, n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        eng.founders(key, E, N, n)
    e1.record(); torch.cuda.synchronize()
    print(f"founders E={E} N={N} n={n}: {e0.elapsed_time(e1)/50*1e3:.1f} us per call (host-inclusive)")
