import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import HipEngine
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
M, E, n = 100_000, 4, 2048
def ev_time(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
gm = synthetic_genetic_map(M, n_chr=21)
eng = HipEngine(gm.recombination_vec, gm.marker_effects)
geno = synthetic_genotypes(n, M)
rng = np.random.default_rng(0)
rows = [eng.upload_population(np.concatenate([geno] * E)), eng.empty_population(E * n)]
idx = [torch.from_numpy(rng.integers(0, n, (E, n, 2), dtype=np.int32)).cuda() for _ in range(4)]
ident = torch.arange(n, dtype=torch.int32).reshape(1, n, 1).expand(E, n, 2).contiguous().cuda()
gebv = eng.empty((E, n, 1), torch.float32)
key = np.array([0, 7], np.uint32); i = [0]
def cross(g, ix=None):
    i[0] += 1
    eng.cross(rows[i[0] & 1], E, n, ix if ix is not None else idx[i[0] % 4], n, key, 0, rows[1 - (i[0] & 1)], g)
for pf in (True, False):
    eng.set_mask_prefetch(pf)
    print(f"prefetch={pf}: cross+GEBV {ev_time(lambda: cross(gebv)):.0f} us, cross only {ev_time(lambda: cross(None)):.0f} us, "
          f"identity parents (no gather randomness) {ev_time(lambda: cross(None, ident)):.0f} us   [4.9 GB algorithmic]")
