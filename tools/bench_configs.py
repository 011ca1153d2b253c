"""Secondary measurements quoted in DESIGN.md (not the bench.py headline): the drop-in call with host
actions and host rewards (PCIe-inclusive), BASELINE configs[2] (2k x 100k, 8192 crosses), standalone GEBV
with 1 and 8 traits on 2k x 100k (configs[4] shape), the reset gather, top-k at configs[3] size."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import VecBreedGym, HipEngine
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes


def timed(fn, reps, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps


# ---- configs[1], the way a reference user calls it: numpy actions in, numpy rewards out, every step synced
gm = synthetic_genetic_map(10_000); geno = synthetic_genotypes(370, 10_000)
env = VecBreedGym(num_envs=16, initial_population=geno, individual_per_gen=256, reward_shaping=True, genetic_map=gm)
env.reset(seed=0)
acts = [np.random.default_rng(i).integers(0, 256, (16, 256, 2)) for i in range(8)]
i = [0]
def drop_in():
    i[0] += 1
    env.step(acts[i[0] % 8])
dt = timed(drop_in, 200)
print(f"configs[1] drop-in call (host int64 actions -> H2D, rewards -> numpy, sync every step): {dt*1e6:.1f} us/step, {4096/dt/1e6:.1f} M crosses/s")

# ---- configs[2]: 2k x 100k, 4 envs x 2048 = 8192 crosses
M = 100_000
gm3 = synthetic_genetic_map(M, n_chr=21); geno3 = synthetic_genotypes(2048, M)
for mode in ("shared", "independent"):
    env3 = VecBreedGym(num_envs=4, initial_population=geno3, individual_per_gen=2048, reward_shaping=True, sync_rewards=False,
                       env_rng=mode, num_generations=1000, genetic_map=gm3)
    env3.reset(seed=0)
    a3 = [torch.from_numpy(np.random.default_rng(i).integers(0, 2048, (4, 2048, 2), dtype=np.int32)).cuda() for i in range(4)]
    j = [0]
    def step3():
        j[0] += 1
        env3.step(a3[j[0] % 4])
    dt = timed(step3, 10, warm=2)
    print(f"configs[2] 2k x 100k, 8192 crosses, env_rng={mode}: {dt*1e3:.2f} ms/step, {8192/dt/1e6:.2f} M crosses/s, {6*M*8192/dt/1e9:.0f} GB/s algorithmic")
    del env3

# ---- standalone GEBV at configs[4] shape: 2k x 100k, T = 1 and 8
for T in (1, 8):
    gmT = synthetic_genetic_map(M, n_chr=21, n_traits=T)
    eng = HipEngine(gmT.recombination_vec, gmT.marker_effects)
    rows = eng.upload_population(geno3)
    out = eng.empty((2048, T), torch.float32)
    dt = timed(lambda: eng.gebv(rows, out), 20)
    print(f"GEBV 2048 x 100k, T={T}: {dt*1e6:.1f} us, {2*M*2048/dt/1e9:.0f} GB/s of genotype bytes, {2048/dt/1e6:.2f} M individuals/s")
    if T == 1:
        idx = torch.from_numpy(np.random.default_rng(0).integers(0, 2048, 8192, dtype=np.int32)).cuda()
        o2 = eng.empty_population(8192); g2 = eng.empty((8192, 1), torch.float32)
        dt = timed(lambda: eng.gather(rows, idx, o2, g2), 10)
        print(f"gather+GEBV 8192 rows x 100k (reset at configs[2]): {dt*1e6:.1f} us, {4*M*8192/dt/1e9:.0f} GB/s read+write")

# ---- top-k at configs[3] size: 65,536 scores, k = 6,553
eng = HipEngine(np.full(64, 0.1, np.float32), np.ones((64, 1), np.float32))
s = torch.randn(65536, device="cuda")
dt = timed(lambda: eng.topk(s, 6553), 50)
print(f"top-k 65,536 -> 6,553: {dt*1e6:.1f} us")
s2 = torch.randn(16, 256, device="cuda")
dt = timed(lambda: eng.topk(s2, 25), 200)
print(f"top-k 16 x 256 -> 25: {dt*1e6:.1f} us")

# ---- configs[3], one rank's share: 8,192 individuals x 100k per GPU, top-k k = 6,553 of the (here: local)
#      population, winners gathered, 8,192 crosses from the replicated pool (collectives are identities at world 1)
from breedgym_amd import Simulator
from breedgym_amd.vector.distributed import ShardedPopulation
gm4 = synthetic_genetic_map(100_000, n_chr=21)
sim = Simulator(genetic_map=gm4, seed=0)
g8 = synthetic_genotypes(2048, 100_000)
rows8 = sim.engine.upload_population(np.concatenate([g8] * 4))
pairs = torch.from_numpy(np.random.default_rng(0).integers(0, 6553, (8192, 2), dtype=np.int32)).cuda()
key = np.array([0, 5], np.uint32)
for storage in ("bytes", "bits"):
    sp = ShardedPopulation(sim, rows8, storage=storage)
    dt = timed(lambda: sp.select_and_cross(6553, pairs, key), 5, warm=2)
    print(f"configs[3] one rank, storage={storage}: GEBV + top-6,553 of 8,192 + gather winners + 8,192 crosses x 100k: "
          f"{dt*1e3:.2f} ms/generation, {8192/dt/1e6:.2f} M crosses/s per GPU")
    del sp
