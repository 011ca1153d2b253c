"""The default env of BASELINE configs[1] and configs[2] with Simulator(mutation=p), p > 0: what the flips cost per step
(mutation_sites_kernel + mutation_apply_kernel + the GEBV / reward passes that then run behind the launch).
usage: python tools/time_mutation.py [c2|c3]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from breedgym_amd import VecBreedGym  # noqa: E402
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes  # noqa: E402

SHAPES = {"c2": (370, 10_000, 16, 256), "c3": (2048, 100_000, 4, 2048)}
for N, M, E, n in [SHAPES[a] for a in sys.argv[1:]] or SHAPES.values():
    gm = synthetic_genetic_map(M, n_chr=10 if M == 10_000 else 21, n_traits=1, seed=1)
    geno = synthetic_genotypes(N, M, seed=0)
    arng = np.random.default_rng(3)
    for p in ((1e-4,) if len(sys.argv) > 1 else (0.0, 1e-4)):
        env = VecBreedGym(num_envs=E, initial_population=geno, individual_per_gen=n, num_generations=10, autoreset=True,
                          reward_shaping=True, sync_rewards=False, genetic_map=gm, mutation=p)
        actions = [torch.from_numpy(arng.integers(0, n, (E, n, 2), dtype=np.int32)).to(env.device) for _ in range(4)]
        env.reset(seed=7)
        dt, _ = bench.time_steps(env, actions, 200, 20, lambda: torch.cuda.synchronize())
        print(f"{N} x {M}, {E} x {n} crosses, mutation {p:g}: {dt / 200 * 1e6:8.1f} us per step", flush=True)
        del env
