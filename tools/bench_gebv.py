import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import HipEngine
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
for rows, M in ((4096, 10_000), (2048, 100_000)):
    gm = synthetic_genetic_map(M, n_traits=8)
    eng = HipEngine(gm.recombination_vec, gm.marker_effects)
    pop = eng.upload_population(synthetic_genotypes(rows, M))
    out = eng.empty((rows, 8), torch.float32)
    for _ in range(20): eng.gebv(pop, out)
    torch.cuda.synchronize()
