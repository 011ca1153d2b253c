"""Pure host cost of VecBreedGym.step: enqueue bursts of 8 steps into an idle queue (the GPU cannot back-pressure
the host within a burst), and a cProfile of the same loop."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import VecBreedGym
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

gm = synthetic_genetic_map(10_000); geno = synthetic_genotypes(370, 10_000)
env = VecBreedGym(num_envs=16, initial_population=geno, individual_per_gen=256, num_generations=1000, reward_shaping=True,
                  sync_rewards=False, genetic_map=gm)
acts = [torch.from_numpy(np.random.default_rng(i).integers(0, 256, (16, 256, 2), dtype=np.int32)).cuda() for i in range(8)]
env.reset(seed=1)
for i in range(30): env.step(acts[i % 8])
torch.cuda.synchronize()
best = 1e9
for trial in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(8): env.step(acts[i % 8])
    t1 = time.perf_counter()
    best = min(best, (t1 - t0) / 8)
print(f"host cost of step(): {best*1e6:.1f} us (best of 20 bursts of 8 into an idle queue)")
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(500): env.step(acts[i % 8])
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"steady state: host loop {1e6*(t1-t0)/500:.1f} us/step, with final sync {1e6*(t2-t0)/500:.1f} us/step")
