"""Can the fixed chain of the headline launch be hidden ACROSS launches?  (~5 us of a ~20 us launch are
dispatch ramp, entry barrier, indices and first words; the rest runs at the chip's store rate.)

One vector env of 16 x 256 (BASELINE configs[1]) against the SAME 16 envs as two halves of eight -- two VecBreedGym, each
with the env window of its half (bh_set_env_window: the founders and keys of envs 0-7 / 8-15 of the 16), each with its
own engine context on its own HIP stream -- stepped (a) one after the other from one host thread, (b) from two host
threads (the C ABI calls release the GIL).  Every half is its own dependency chain, so half A's fixed chain can run under
half B's stores.  Prints crosses/s of each arrangement; a probe, not a bench line.

usage: python tools/probe_overlap.py [steps]
"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes  # noqa: E402
from breedgym_amd.vector import VecBreedGym  # noqa: E402

N, M, E, n, G = 370, 10_000, 16, 256, 10
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


def make(num_envs, first, total, gm, geno):
    env = VecBreedGym(num_envs=num_envs, initial_population=geno, individual_per_gen=n, num_generations=G, autoreset=True,
                      reward_shaping=True, env_rng="shared", sync_rewards=False, genetic_map=gm)
    env.simulator.engine.set_env_window(first, total)
    return env


def actions_for(num_envs, seed, dev):
    r = np.random.default_rng(seed)
    return [torch.from_numpy(r.integers(0, n, (num_envs, n, 2), dtype=np.int32)).to(dev) for _ in range(16)]


def settle(step, seconds=0.2):
    t_end, i = time.perf_counter() + seconds, 0
    while time.perf_counter() < t_end:
        step(i)
        i += 1
        if i % 64 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()


def main():
    gm = synthetic_genetic_map(M, n_chr=10, n_traits=1, seed=1)
    geno = synthetic_genotypes(N, M, seed=0)

    # ---- the line's arrangement: one env of 16, one stream
    env = make(E, 0, E, gm, geno)
    acts = actions_for(E, 2, env.device)
    env.reset(seed=100)
    settle(lambda i: env.step(acts[i % 16]))
    t0 = time.perf_counter()
    for i in range(STEPS):
        env.step(acts[i % 16])
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"one env of 16, one stream         : {dt / STEPS * 1e6:7.2f} us per step (host issue {t_issue / STEPS * 1e6:5.2f}), "
          f"{E * n * STEPS / dt / 1e6:6.1f} M crosses/s")
    del env

    # ---- two halves, two contexts, two streams
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    halves, hacts = [], []
    for h, s in enumerate(streams):
        with torch.cuda.stream(s):
            e = make(E // 2, h * (E // 2), E, gm, geno)
            e.simulator.engine.set_stream(s)
            e.reset(seed=100)
            halves.append(e)
            hacts.append(actions_for(E // 2, 2 + h, e.device))
    torch.cuda.synchronize()

    def step_half(h, i):
        with torch.cuda.stream(streams[h]):
            halves[h].step(hacts[h][i % 16])

    def both(i):
        step_half(0, i)
        step_half(1, i)

    settle(both)
    t0 = time.perf_counter()
    for i in range(STEPS):
        both(i)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"two halves of 8, one host thread  : {dt / STEPS * 1e6:7.2f} us per step of both (host issue {t_issue / STEPS * 1e6:5.2f}), "
          f"{E * n * STEPS / dt / 1e6:6.1f} M crosses/s")

    # one half alone: what a launch of half the crosses takes
    settle(lambda i: step_half(0, i))
    t0 = time.perf_counter()
    for i in range(STEPS):
        step_half(0, i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"one half of 8 alone               : {dt / STEPS * 1e6:7.2f} us per step, {E // 2 * n * STEPS / dt / 1e6:6.1f} M crosses/s")

    # two host threads, one per half
    settle(both)
    go = threading.Barrier(3)

    def worker(h):
        go.wait()
        for i in range(STEPS):
            step_half(h, i)
        streams[h].synchronize()

    th = [threading.Thread(target=worker, args=(h,)) for h in (0, 1)]
    for t in th:
        t.start()
    go.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print(f"two halves of 8, two host threads : {dt / STEPS * 1e6:7.2f} us per step of both, {E * n * STEPS / dt / 1e6:6.1f} M crosses/s")


if __name__ == "__main__":
    main()
