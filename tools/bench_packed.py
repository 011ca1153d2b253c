"""Timing of bh_cross_packed against bh_cross: BASELINE configs[1] (16 x 256 crosses x 10k markers) and
configs[2] (4 x 2048 x 100k), both RNG modes, with and without the byte-layout observation."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import HipEngine
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes


def bench(M, E, n, N, n_chr, reps):
    gm = synthetic_genetic_map(M, n_chr=n_chr)
    eng = HipEngine(gm.recombination_vec, gm.marker_effects)
    geno = synthetic_genotypes(N, M)
    rng = np.random.default_rng(0)
    sel = np.concatenate([rng.permutation(N)[:n] if n <= N else rng.integers(0, N, n) for _ in range(E)])
    rows = [eng.upload_population(geno[sel]), eng.empty_population(E * n)]
    prow = [eng.pack(rows[0]), eng.empty_packed(E * n)]
    obs = eng.empty_population(E * n)
    idx = [torch.from_numpy(rng.integers(0, n, (E, n, 2), dtype=np.int32)).cuda() for _ in range(8)]
    gebv = eng.empty((E, n, 1), torch.float32)
    rew = eng.empty((E,), torch.float32)

    def timed(step):
        key = np.array([0, 7], np.uint32)
        for i in range(6):
            step(i, key)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            step(i, key)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    print(f"--- {E} x {n} crosses x {M} markers")
    for mode in (0, 1):
        name = "independent" if mode else "shared"
        us = timed(lambda i, key: eng.cross(rows[i & 1], E, n, idx[i % 8], n, key, mode, rows[1 - (i & 1)], gebv, rew))
        print(f"{name:11s} byte layout                 : {us:8.1f} us/step  {E*n/us:6.1f} M crosses/s")
        us = timed(lambda i, key: eng.cross_packed(prow[i & 1], E, n, idx[i % 8], n, key, mode, prow[1 - (i & 1)], obs,
                                                   gebv, rew))
        print(f"{name:11s} packed + byte observation   : {us:8.1f} us/step  {E*n/us:6.1f} M crosses/s")
        us = timed(lambda i, key: eng.cross_packed(prow[i & 1], E, n, idx[i % 8], n, key, mode, prow[1 - (i & 1)], None,
                                                   gebv, rew))
        print(f"{name:11s} packed, no observation      : {us:8.1f} us/step  {E*n/us:6.1f} M crosses/s")
        us = timed(lambda i, key: eng.cross_packed(prow[i & 1], E, n, idx[i % 8], n, key, mode, prow[1 - (i & 1)], None,
                                                   None, None))
        print(f"{name:11s} packed, no observation/GEBV : {us:8.1f} us/step  {E*n/us:6.1f} M crosses/s")
    us = timed(lambda i, key: eng.unpack(prow[0], obs))
    print(f"unpack {E*n} rows                        : {us:8.1f} us")
    us = timed(lambda i, key: eng.pack(rows[0], prow[1]))
    print(f"pack {E*n} rows                          : {us:8.1f} us")
    us = timed(lambda i, key: eng.gather_packed(prow[0], None, E * n, None, None, gebv))
    print(f"GEBV of {E*n} packed rows                : {us:8.1f} us")


bench(10_000, 16, 256, 370, 10, 200)
bench(100_000, 4, 2048, 2000, 21, 20)
