"""Timing of bh_cross_packed (bit planes in, bit planes + byte observation + GEBV + reward out) at 10k markers for growing
batches: where the observation stops fitting the Infinity Cache.  One line per shape: us per launch, by HIP events."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import HipEngine
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes


def bench(M, E, n, N, reps):
    gm = synthetic_genetic_map(M, n_chr=10)
    eng = HipEngine(gm.recombination_vec, gm.marker_effects)
    geno = synthetic_genotypes(N, M)
    rng = np.random.default_rng(0)
    sel = rng.integers(0, N, E * n)
    prow = [eng.pack(eng.upload_population(geno[sel])), eng.empty_packed(E * n)]
    obs = [eng.empty_population(E * n), eng.empty_population(E * n)]
    idx = [torch.from_numpy(rng.integers(0, n, (E, n, 2), dtype=np.int32)).cuda() for _ in range(4)]
    gebv = eng.empty((E, n, 1), torch.float32)
    rew = eng.empty((E,), torch.float32)
    key = np.array([0, 7], np.uint32)

    def step(i):
        eng.cross_packed(prow[i & 1], E, n, idx[i % 4], n, key, 0, prow[1 - (i & 1)], obs[i & 1], gebv, rew)

    for i in range(6):
        step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        step(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    mb = E * n * (2.75 * M) / 1e6
    print(f"{E:4d} x {n:5d} x {M}: {us:9.1f} us per launch  {mb / us * 1e-6 * 1e6 / 1e3:6.2f} TB/s on {mb:7.0f} MB")


for E, n, reps in ((16, 256, 400), (32, 256, 300), (64, 256, 200), (64, 512, 100), (64, 1024, 50)):
    bench(10_000, E, n, 370, reps)
