"""The reference's own timing protocol (scripts/time_vec.py:47-71): mean seconds per
``env.step(np.random.randint(0, 210, size=(E, 210, 2)))`` over 10 repeats after a reset, for
E in {8, 16, 32}, n = 210, M = 10k -- host-generated actions inside the timed call, as there."""
import sys, timeit
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import VecBreedGym
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
from breedgym_amd.vector import SelectionScores
from breedgym_amd.vector.breeding_programs_env import WheatBreedGym

gm = synthetic_genetic_map(10_000); geno = synthetic_genotypes(370, 10_000)
print("E     s/step (mean of 10, synced)   crosses/s")
for E in (8, 16, 32):
    env = VecBreedGym(num_envs=E, initial_population=geno, individual_per_gen=210, genetic_map=gm)
    env.reset(seed=0)
    for _ in range(11):  # a whole episode and its autoreset: every kernel of the loop has been loaded once (the first
        env.step(np.random.randint(0, 210, size=(E, 210, 2)))  # use of a kernel costs ~50 ms of code-object loading)
    def one():
        env.step(np.random.randint(0, 210, size=(E, 210, 2)))
        torch.cuda.synchronize()
    t = timeit.timeit(one, number=10) / 10
    print(f"{E:<5d} {t:.6f}                      {E*210/t/1e6:.1f} M")

# the policy loops of the reference's seeded tests (tests/test_vec.py:136-154): GEBV scores -> SelectionScores
env = SelectionScores(k=10, num_envs=16, initial_population=geno, individual_per_gen=210, genetic_map=gm,
                      reward_shaping=True, sync_rewards=False, num_generations=10**6)
_, infos = env.reset(seed=7)
for _ in range(5): _, _, _, _, infos = env.step(infos["GEBV"][..., 0])
torch.cuda.synchronize(); t0 = timeit.default_timer()
for _ in range(200): _, _, _, _, infos = env.step(infos["GEBV"][..., 0])
torch.cuda.synchronize(); t = (timeit.default_timer() - t0) / 200
print(f"SelectionScores(k=10) GEBV policy, 16 x 210, scores stay on device: {t*1e6:.1f} us/step, {16*210/t/1e6:.1f} M crosses/s")

w = WheatBreedGym(VecBreedGym(num_envs=4, initial_population=geno, individual_per_gen=200, genetic_map=gm, sync_rewards=False,
                              num_generations=10**6), n_lines=200, plant_per_line=100, k_per_line=5)
w.reset(seed=0)
sc = torch.randn(4, 200, 200, device="cuda")
for _ in range(2): w.step(sc)
torch.cuda.synchronize(); t0 = timeit.default_timer()
for _ in range(10): w.step(sc)
torch.cuda.synchronize(); t = (timeit.default_timer() - t0) / 10
print(f"WheatBreedGym 4 envs x (200 crosses + 20,000 doubled haploids + 2 selections) x 10k markers: {t*1e3:.2f} ms/step")

# BASELINE configs[0]: single env, 370 x 10k, 16 random crosses per generation, 10 generations
from breedgym_amd import BreedGym
env = BreedGym(initial_population=geno, genetic_map=gm)
env.reset(seed=0)
def episode():
    env.reset()
    n = 370
    for _ in range(10):
        env.step(np.random.randint(0, n, size=(16, 2)))
        n = 16
episode(); torch.cuda.synchronize()
t = timeit.timeit(episode, number=20) / 20
print(f"configs[0] BreedGym, 16 crosses/gen x 10 generations (info['GEBV'] as a DataFrame every step): {t*1e3:.2f} ms/episode, {t/10*1e6:.0f} us/step")
