"""The draws of reset() and SelectionScores.step() on the device: bh_founders_device and bh_selection_actions, from the
sizes one workgroup does in LDS to the ones that take the radix sort in HBM (csrc/sort_kernels.h)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import HipEngine

eng = HipEngine(np.full(64, 0.01, np.float32), np.ones((64, 1), np.float32))
key = np.array([0, 3], np.uint32)


def timed(f, reps):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for N, n, E in ((370, 256, 16), (2000, 2000, 4), (8192, 819, 8), (8193, 819, 8), (65536, 8192, 1), (65536, 65536, 4),
                (1 << 20, 4096, 1), (1 << 20, 1 << 20, 1)):
    print(f"founders E={E} N={N} n={n}: {timed(lambda: eng.founders(key, E, N, n), 20):.1f} us per call (host-inclusive)", flush=True)
for E, n, k, nc in ((16, 256, 25, 256), (4, 2048, 204, 2048), (1, 8192, 819, 8192), (1, 65536, 6553, 8192)):
    s = torch.randn((E, n), device=eng.device)
    print(f"selection_actions E={E} n={n} k={k} ({k * (k - 1) // 2} pairs) n_crosses={nc}: "
          f"{timed(lambda: eng.selection_actions(s, k, nc, key), 10):.1f} us per call (host-inclusive)", flush=True)
