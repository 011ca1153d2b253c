"""The reference's own cross timing (scripts/time_cross.py:11-42): `simulator.cross(parents)` on materialised parents
of 300 progenies x 1,406,757 markers (all-zero genotypes, as the script makes them), mean of 25 calls after one
warm-up, each call synchronised -- through breedgym_amd.Simulator.cross.  The reference's genetic_map.txt for that
shape is not shipped: a synthetic map of the same length stands in."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from breedgym_amd import Simulator
from breedgym_amd.synthetic import synthetic_genetic_map

for n_prog, M in ((300, 10_000), (300, 150_000), (300, 1_406_757)):
    gm = synthetic_genetic_map(M, n_chr=21, seed=1)
    sim = Simulator(genetic_map=gm, seed=0)
    parents = torch.zeros((n_prog, 2, M, 2), dtype=torch.bool, device=sim.engine.device)
    sim.cross(parents)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(25):
        out = sim.cross(parents)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 25
    print(f"{n_prog} progenies x {M} markers: {dt * 1e3:.3f} ms per simulator.cross(parents) "
          f"({n_prog / dt:.0f} crosses/s, {6 * M * n_prog / dt / 1e9:.0f} GB/s at the API's 6 M bytes per cross)", flush=True)
    del sim, parents, out
    torch.cuda.empty_cache()
