"""Timing of the env step at 16 x 256 x 10k with one and two traits (the two-trait stream holds twice the digit registers)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from breedgym_amd import VecBreedGym
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
geno = synthetic_genotypes(370, 10_000, seed=0)
for T in (1, 2):
    gm = synthetic_genetic_map(10_000, n_chr=10, n_traits=T, seed=1)
    env = VecBreedGym(num_envs=16, initial_population=geno, individual_per_gen=256, num_generations=10, reward_shaping=True,
                      genetic_map=gm, sync_rewards=False)
    env.reset(seed=1)
    acts = [torch.from_numpy(np.random.default_rng(i).integers(0, 256, (16, 256, 2), dtype=np.int32)).cuda() for i in range(8)]
    for i in range(300): env.step(acts[i % 8])
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(3000): env.step(acts[i % 8])
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{T} trait(s): {dt / 3000 * 1e6:.2f} us per step")
