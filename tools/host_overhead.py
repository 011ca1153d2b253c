"""How long does the host take to enqueue one VecBreedGym.step? (must stay below the GPU step time)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import VecBreedGym
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
import cProfile, pstats

gm = synthetic_genetic_map(10_000); geno = synthetic_genotypes(370, 10_000)
env = VecBreedGym(num_envs=16, initial_population=geno, individual_per_gen=256, num_generations=10, reward_shaping=True,
                  sync_rewards=False, genetic_map=gm)
acts = [torch.from_numpy(np.random.default_rng(i).integers(0, 256, (16, 256, 2), dtype=np.int32)).cuda() for i in range(8)]
env.reset(seed=1)
for i in range(30): env.step(acts[i % 8])
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    for i in range(200): env.step(acts[i % 8])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue {1e6*(t1-t0)/200:.1f} us/step, total {1e6*(t2-t0)/200:.1f} us/step")
pr = cProfile.Profile(); pr.enable()
for i in range(200): env.step(acts[i % 8])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
