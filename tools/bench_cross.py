"""Quick A/B timing of bh_cross at BASELINE configs[1] size: prefetch on/off, both RNG modes."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import HipEngine
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

M, E, n = 10_000, 16, 256
gm = synthetic_genetic_map(M)
eng = HipEngine(gm.recombination_vec, gm.marker_effects)
geno = synthetic_genotypes(370, M)
rng = np.random.default_rng(0)
rows = [eng.upload_population(geno[np.concatenate([rng.permutation(370)[:n] for _ in range(E)])]), eng.empty_population(E * n)]
idx = [torch.from_numpy(rng.integers(0, n, (E, n, 2), dtype=np.int32)).cuda() for _ in range(8)]
gebv = eng.empty((E, n, 1), torch.float32)

def run(mode, prefetch, reps=200):
    eng.set_mask_prefetch(prefetch)
    key = np.array([0, 7], np.uint32)
    for i in range(10):
        eng.cross(rows[i & 1], E, n, idx[i % 8], n, key, mode, rows[1 - (i & 1)], gebv)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        eng.cross(rows[i & 1], E, n, idx[i % 8], n, key, mode, rows[1 - (i & 1)], gebv)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"mode={'independent' if mode else 'shared':11s} prefetch={prefetch!s:5s}: {us:7.1f} us/step  {E*n/us:6.1f} M crosses/s  {6*M*E*n/us/1e3:7.0f} GB/s")

for rep in range(2):
    for mode in (1, 0):
        for pf in (True, False):
            run(mode, pf)

# host-side cost of one call (how fast can the CPU enqueue steps?)
import time
for pf in (True, False):
    eng.set_mask_prefetch(pf)
    key = np.array([0, 7], np.uint32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(50):
        eng.cross(rows[i & 1], E, n, idx[i % 8], n, key, 0, rows[1 - (i & 1)], gebv)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue per call prefetch={pf}: {(t1-t0)/50*1e6:.1f} us (drain {(t2-t1)*1e6:.0f} us)")
