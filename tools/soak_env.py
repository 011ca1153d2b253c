"""Soak of the env itself: VecBreedGym at BASELINE configs[1] (16 x 256 x 10k, autoreset, defaults) over many episodes
-- every step's reward and GEBV against the CPU oracle's chain, the observation every `check` steps and at every
horizon (the founders the horizon step's launch gathered ahead).  The paths of round 4's second session in one loop:
write-through observation stores, the LDS digit copy, founders drawn an episode ahead, the reset made by the horizon
launch."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import VecBreedGym
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
from oracle import c_oracle as co

N, M, E, n, G = 370, 10_000, 16, 256, 10
episodes, check = int(sys.argv[1]) if len(sys.argv) > 1 else 12, 4
gm = synthetic_genetic_map(M, n_chr=10, n_traits=1, seed=1)
geno = synthetic_genotypes(N, M, seed=0)
env = VecBreedGym(num_envs=E, initial_population=geno, individual_per_gen=n, num_generations=G, reward_shaping=True, genetic_map=gm)
co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
pop, info = env.reset(seed=42)
f, env_key = co.founders(co.key_from_seed(42), E, N, n)
ref = geno.view(np.uint8)[f]
sim_key = co.key_from_seed(42)
rng = np.random.default_rng(7)
bad = 0
u8 = lambda x: np.asarray(x).view(np.uint8).reshape(E, n, M, 2)
for step in range(episodes * G):
    a = rng.integers(0, n, (E, n, 2), dtype=np.int32)
    pop, rews, ter, tru, info = env.step(a)
    ref, sim_key = co.cross(ref, a, gm.recombination_vec, sim_key, 0)
    g64 = co.gebv(ref, gm.marker_effects, want64=True)[1]
    ok = np.all(np.abs(np.asarray(info["GEBV"]).astype(np.float64) - g64) <= 1e-5 * np.maximum(np.abs(g64), 1.0))
    ok &= np.allclose(np.asarray(rews), g64.max(axis=(1, 2)), rtol=1e-5) and np.array_equal(env.simulator.random_key, sim_key)
    horizon = (step + 1) % G == 0
    if horizon:
        f, env_key = co.founders(env_key, E, N, n)
        ref = geno.view(np.uint8)[f]
        ok &= bool(tru.all()) and np.array_equal(env.random_key, env_key)
    if horizon or step % check == 0:
        ok &= np.array_equal(u8(pop), ref)
    if not ok:
        bad += 1
        print(f"  MISMATCH at step {step}")
print(f"{episodes} episodes of {G} generations, {E} x {n} x {M}: {bad} mismatches")
sys.exit(1 if bad else 0)
