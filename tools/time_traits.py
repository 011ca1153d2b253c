"""The default env of BASELINE configs[1] (16 x 256 x 10k) with 1 .. 8 traits: per step, HIP-synchronised wall time.
usage: python tools/time_traits.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from breedgym_amd import VecBreedGym  # noqa: E402
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes  # noqa: E402

N, M, E, n = 370, 10_000, 16, 256
geno = synthetic_genotypes(N, M, seed=0)
arng = np.random.default_rng(3)
for T in (1, 2, 3, 4, 8):
    gm = synthetic_genetic_map(M, n_chr=10, n_traits=T, seed=1)
    for obs in ("eager", "lazy"):
        env = VecBreedGym(num_envs=E, initial_population=geno, individual_per_gen=n, num_generations=10, autoreset=True,
                          reward_shaping=True, sync_rewards=False, genetic_map=gm, observations=obs)
        actions = [torch.from_numpy(arng.integers(0, n, (E, n, 2), dtype=np.int32)).to(env.device) for _ in range(4)]
        env.reset(seed=7)
        dt, _ = bench.time_steps(env, actions, 400, 20, lambda: torch.cuda.synchronize())
        print(f"{T} traits, observations {obs}: {dt / 400 * 1e6:7.1f} us per step", flush=True)
        del env
