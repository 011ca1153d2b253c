"""One variant of the configs[1] cross per process, for rocprofv3 --kernel-trace --stats (the Python loop
itself is launch-bound at this size, so only the profiler's kernel durations mean anything).
usage: prof_packed.py {bytes|packed_obs|packed|packed_nogebv|unpack|pgebv} [mode]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import HipEngine
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

variant = sys.argv[1]
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
M, E, n, N = 10_000, 16, 256, 370
gm = synthetic_genetic_map(M)
eng = HipEngine(gm.recombination_vec, gm.marker_effects)
geno = synthetic_genotypes(N, M)
rng = np.random.default_rng(0)
sel = np.concatenate([rng.permutation(N)[:n] for _ in range(E)])
rows = [eng.upload_population(geno[sel]), eng.empty_population(E * n)]
prow = [eng.pack(rows[0]), eng.empty_packed(E * n)]
obs = eng.empty_population(E * n)
idx = [torch.from_numpy(rng.integers(0, n, (E, n, 2), dtype=np.int32)).cuda() for _ in range(8)]
gebv = eng.empty((E, n, 1), torch.float32)
key = np.array([0, 7], np.uint32)
for i in range(100):
    a, b = i & 1, 1 - (i & 1)
    if variant == "bytes":
        eng.cross(rows[a], E, n, idx[i % 8], n, key, mode, rows[b], gebv)
    elif variant == "packed_obs":
        eng.cross_packed(prow[a], E, n, idx[i % 8], n, key, mode, prow[b], obs, gebv)
    elif variant == "packed":
        eng.cross_packed(prow[a], E, n, idx[i % 8], n, key, mode, prow[b], None, gebv)
    elif variant == "packed_nogebv":
        eng.cross_packed(prow[a], E, n, idx[i % 8], n, key, mode, prow[b], None, None)
    elif variant == "unpack":
        eng.unpack(prow[0], obs)
    elif variant == "pgebv":
        eng.gather_packed(prow[0], None, E * n, None, None, gebv)
torch.cuda.synchronize()
