"""Row-role experiments at BASELINE configs[1] size (shared keys): with / without GEBV, and the standalone ops."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import HipEngine
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

M, E, n = 10_000, 16, 256
def ev_time(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3

for T in (1, 2, 3, 4, 8):
    gm = synthetic_genetic_map(M, n_traits=T)
    eng = HipEngine(gm.recombination_vec, gm.marker_effects)
    geno = synthetic_genotypes(370, M)
    rng = np.random.default_rng(0)
    rows = [eng.upload_population(geno[np.concatenate([rng.permutation(370)[:n] for _ in range(E)])]), eng.empty_population(E * n)]
    idx = [torch.from_numpy(rng.integers(0, n, (E, n, 2), dtype=np.int32)).cuda() for _ in range(8)]
    gebv = eng.empty((E, n, T), torch.float32)
    key = np.array([0, 7], np.uint32)
    i = [0]
    def cross(g):
        i[0] += 1
        eng.cross(rows[i[0] & 1], E, n, idx[i[0] % 8], n, key, 0, rows[1 - (i[0] & 1)], g)
    rew = eng.empty((E,), torch.float32)
    def cross_r():
        i[0] += 1
        eng.cross(rows[i[0] & 1], E, n, idx[i[0] % 8], n, key, 0, rows[1 - (i[0] & 1)], gebv, rew)
    def cross_sep():
        cross(gebv); eng.reward_max(gebv, E, rew)
    t_r = ev_time(cross_r); t_s = ev_time(cross_sep)
    print(f"T={T}: cross+GEBV+fused reward {t_r:.1f} us; cross+GEBV then reward kernel {t_s:.1f} us")
    t_g = ev_time(lambda: cross(gebv)); t_n = ev_time(lambda: cross(None))
    t_gebv = ev_time(lambda: eng.gebv(rows[0], gebv.reshape(E * n, T)))
    print(f"T={T}: cross+GEBV {t_g:.1f} us, cross only {t_n:.1f} us, standalone GEBV of 4096 rows {t_gebv:.1f} us "
          f"({2*M*E*n/t_gebv/1e3:.0f} GB/s)")
