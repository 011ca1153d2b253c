"""What the reference's timing protocol (scripts/time_vec.py:57-65: env.step(np.random.randint(...)) with a sync per step)
is made of at E = 16, n = 210, M = 10k: NumPy's randint, the upload of host actions, the step, the sync."""
import sys, timeit
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import VecBreedGym
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
gm = synthetic_genetic_map(10_000); geno = synthetic_genotypes(370, 10_000)
E = 16
env = VecBreedGym(num_envs=E, initial_population=geno, individual_per_gen=210, genetic_map=gm)
env.reset(seed=0)
host = [np.random.randint(0, 210, size=(E, 210, 2)) for _ in range(16)]
host32 = [h.astype(np.int32) for h in host]
dev = [torch.from_numpy(h).cuda() for h in host32]
for i in range(30): env.step(host[i % 16])
torch.cuda.synchronize()
def run(f, n=200):
    i = [0]
    def one():
        f(i[0]); i[0] += 1
        torch.cuda.synchronize()
    return timeit.timeit(one, number=n) / n * 1e6
print("randint alone            %.1f us" % (timeit.timeit(lambda: np.random.randint(0, 210, size=(E, 210, 2)), number=200) / 200 * 1e6))
print("sync only                %.1f us" % (timeit.timeit(lambda: torch.cuda.synchronize(), number=200) / 200 * 1e6))
print("device actions + sync    %.1f us" % run(lambda i: env.step(dev[i % 16])))
print("host int32 actions + sync %.1f us" % run(lambda i: env.step(host32[i % 16])))
print("host int64 actions + sync %.1f us" % run(lambda i: env.step(host[i % 16])))
print("randint + step + sync    %.1f us" % run(lambda i: env.step(np.random.randint(0, 210, size=(E, 210, 2)))))
