"""bh_topk at the shapes where its routes meet (chunk sorts in LDS, radix select, the radix sort in HBM)."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from breedgym_amd import HipEngine
eng = HipEngine(np.full(64, 0.01, np.float32), np.ones((64, 1), np.float32))
def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for E, n, k in ((1, 65536, 6553), (4, 65536, 6553), (1, 49152, 4915), (1, 40000, 4000), (8, 100000, 5000), (1, 1 << 20, 10000),
                (2, 20000, 1000), (1, 20000, 2500), (4, 40000, 200), (1, 8192, 819), (16, 256, 25)):
    s = torch.randn((E, n), device=eng.device)
    print(f"topk E={E} n={n} k={k}: {timed(lambda: eng.topk(s, k)):.1f} us", flush=True)
