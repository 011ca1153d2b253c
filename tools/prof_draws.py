"""Two draws for a kernel-level profile (rocprofv3 --kernel-trace --stats -- python3 tools/prof_draws.py [which]):
bh_founders_device at (1, 65536, 8192) and bh_selection_actions at (1, 65536, k = 6553, 8192 crosses)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import HipEngine

which = sys.argv[1] if len(sys.argv) > 1 else "both"
eng = HipEngine(np.full(64, 0.01, np.float32), np.ones((64, 1), np.float32))
key = np.array([0, 3], np.uint32)
if which in ("both", "founders"):
    for _ in range(20):
        eng.founders(key, 1, 65536, 8192)
if which in ("both", "selection"):
    s = torch.randn((1, 65536), device=eng.device)
    for _ in range(10):
        eng.selection_actions(s, 6553, 8192, key)
torch.cuda.synchronize()
