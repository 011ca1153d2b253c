#!/usr/bin/env python
"""bench.py -- crosses/s through vector env.step() on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (N = 1): BASELINE.json configs[1] -- a 370-individual x 10k-marker synthetic germplasm,
16 envs x 256 individuals = 4096 crosses per generation, GEBV reward every step, episodes of 10
generations with autoreset (so every 10th step also pays a reset, as in the reference's
scripts/time_vec.py:57-65 protocol).  A "step" is one ``VecBreedGym.step``: fused parent
gather + meiosis + offspring write + GEBV, the reward reduction, and the reset when due.
Inputs (germplasm, genetic map, actions) are resident in HBM before the timed region; outputs
stay on the device (``sync_rewards=False``), nothing is copied to the host inside it.

N > 1: one process per GPU, every rank steps its own 16 envs (env-sharded): there
is no data-path collective, so scaling is "weak"; value = all ranks' crosses / max rank time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

N_GERMPLASM, N_MARKERS, N_ENVS, N_PER_ENV, N_GENERATIONS = 370, 10_000, 16, 256, 10
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(gm, geno, rng_mode, seconds_hint=8.0):
    """The CPU oracle (a from-scratch port: the reference's chromax/JAX CPU path cannot be
    installed here or on the GPU box) timed on this host on a bounded sample of the same
    workload: whole 4096-cross generations of 370x10k (cross + GEBV + reward), OpenMP over
    crosses on the box's CPU share (16 hardware threads per GPU)."""
    from oracle import c_oracle as co
    from oracle import numpy_oracle as no

    rng = np.random.default_rng(0)
    E, n, M = N_ENVS, N_PER_ENV, N_MARKERS
    pool = geno.view(np.uint8)[np.stack([rng.permutation(N_GERMPLASM)[:n] for _ in range(E)])]
    idx = rng.integers(0, n, (E, n, 2), dtype=np.int32)
    key = co.key_from_seed(0)
    cores = min(16, len(os.sched_getaffinity(0)))
    co.set_num_threads(cores)

    def one_generation():
        out, _ = co.cross(pool, idx, gm.recombination_vec, key, rng_mode)
        g = co.gebv(out, gm.marker_effects)
        co.reward_max(g)

    t0 = time.perf_counter()
    one_generation()  # also the warm-up
    first = time.perf_counter() - t0
    reps = int(max(1, min(50, seconds_hint / max(first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        one_generation()
    dt = (time.perf_counter() - t0) / reps
    res = {
        "value": E * n / dt, "unit": "crosses/s", "cores": cores, "kind": "port",
        "sample": f"{reps} generations of {E}x{n}={E*n} crosses x {M} markers (C oracle, OpenMP, "
                  f"cross+GEBV+reward, same RNG sharing as the GPU run)",
    }
    # single-thread C and the unfused NumPy restatement on one env's 256 crosses
    co.set_num_threads(1)
    t0 = time.perf_counter()
    out1, _ = co.cross(pool[:1], idx[:1], gm.recombination_vec, key, 1)
    co.gebv(out1, gm.marker_effects)
    res["c_oracle_1thread_crosses_per_s"] = n / (time.perf_counter() - t0)
    co.set_num_threads(cores)
    par = pool[0][idx[0]]  # materialised (n, 2, M, 2), as the reference does
    t0 = time.perf_counter()
    o, _ = no.cross_batched(par, gm.recombination_vec, key)
    no.gebv(o, gm.marker_effects)
    res["numpy_restatement_crosses_per_s"] = n / (time.perf_counter() - t0)
    return res


def time_steps(env, actions, steps, warmup, barrier):
    import torch

    for i in range(warmup):
        env.step(actions[i % len(actions)])
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(actions[i % len(actions)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    barrier()
    return dt


def time_cross_launches(env, actions, reps):
    """HIP events around `reps` back-to-back launches of the hot-path kernel alone (the fused two-role
    launch: streams this generation, makes the next one's masks), in the env's own storage layout."""
    import torch

    eng = env.simulator.engine
    E, n = env.num_envs, env._n
    gebv = eng.empty((E, n, eng.n_traits), torch.float32)
    key = np.ascontiguousarray(env.simulator.random_key.copy())
    if env.storage == "bits":
        pool, outp = env._prows, eng.empty_packed(E * n)
        obs = eng.empty_population(E * n) if env.observations == "eager" else None

        def launch(i):
            eng.cross_packed(pool, E, n, actions[i % len(actions)], n, key, env._rng_mode, outp, obs, gebv)
    else:
        pool, out = env._rows, eng.empty_population(E * n)

        def launch(i):
            eng.cross(pool, E, n, actions[i % len(actions)], n, key, env._rng_mode, out, gebv)

    for i in range(5):
        launch(i)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for i in range(reps):
        launch(i)
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (profiling runs)")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline configuration: no CPU leg, no other key mode / storage lines (profiling runs, "
                         "so that a kernel name in the profile is one configuration)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")

    import torch
    import torch.distributed as dist

    from breedgym_amd import VecBreedGym
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: breedgym_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run (also with one rank)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    gm = synthetic_genetic_map(N_MARKERS, n_chr=10, n_traits=1, seed=1)
    geno = synthetic_genotypes(N_GERMPLASM, N_MARKERS, seed=0)

    def make_env(env_rng, **kw):
        return VecBreedGym(
            num_envs=N_ENVS, initial_population=geno, individual_per_gen=N_PER_ENV, num_generations=N_GENERATIONS,
            autoreset=True, reward_shaping=True, env_rng=env_rng, sync_rewards=False, genetic_map=gm, **kw,
        )

    # Headline: the drop-in configuration, all defaults.  env_rng="shared" is what the reference
    # computes -- its jax.vmap over an un-batched simulator key hands every env the same gamete keys
    # (vec_env.py:75-77) -- so per step the Threefry work is n x 2 gametes, applied to E x n crosses.
    # storage="bits" (default): the populations live as bit planes between steps; every step still
    # writes and returns the byte-layout observation (observations="eager").
    env = make_env("shared")
    dev = env.device
    arng = np.random.default_rng(2 + rank)
    actions = [
        torch.from_numpy(arng.integers(0, N_PER_ENV, (N_ENVS, N_PER_ENV, 2), dtype=np.int32)).to(dev)
        for _ in range(16)
    ]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    env.reset(seed=100 + rank)
    dt = time_steps(env, actions, args.steps, args.warmup, barrier)
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())
    crosses_per_step = N_ENVS * N_PER_ENV
    kern_ms = time_cross_launches(env, actions, max(20, args.steps))

    # The same steps with an independent Threefry stream per cross (an extension: not what the
    # reference computes).  VALU-bound: 2 x 10k x 36 integer ops per cross.
    ti = kern_ms_i = None
    if not args.headline_only:
        env_i = make_env("independent")
        env_i.reset(seed=100 + rank)
        dt_i = time_steps(env_i, actions, args.steps, args.warmup, barrier)
        ti = torch.tensor([dt_i], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(ti, op=dist.ReduceOp.MAX)
        kern_ms_i = time_cross_launches(env_i, actions, max(20, args.steps))

    # The other storage / observation choices of the same env, same outputs (one GPU only: extra
    # lines, not the headline): byte-layout state, and bit planes with on-demand observations.
    variants = {}
    if world == 1 and not args.headline_only:
        for name, kw in (("byte_layout_storage", dict(storage="bytes")),
                         ("lazy_observations", dict(storage="bits", observations="lazy"))):
            env_v = make_env("shared", **kw)
            env_v.reset(seed=100 + rank)
            dt_v = time_steps(env_v, actions, args.steps, args.warmup, barrier)
            variants[name] = {"value": args.steps * N_ENVS * N_PER_ENV / dt_v, "ms_per_step": dt_v / args.steps * 1e3,
                              "kernel_ms": time_cross_launches(env_v, actions, max(20, args.steps))}
            del env_v

    # algorithmic bytes per launch: 2 parents x 2M read + 2M written per cross
    alg_bytes = 6 * N_MARKERS * crosses_per_step
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic = None
    pmc = ROOT / "profiles" / "pmc_traffic.json"
    if pmc.exists():
        try:
            traffic = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        line = {
            "metric": "crosses/sec (vector env.step) at 370 pop x 10k markers",
            "value": world * args.steps * crosses_per_step / dt_max,
            "unit": "crosses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: 370x10k germplasm, 16 envs x 256 = 4096 crosses/gen, GEBV reward "
                            "every step, 10-generation episodes with autoreset",
                "n_markers": N_MARKERS, "crosses_per_step_per_gpu": crosses_per_step,
                "env_rng": "shared (reference semantics: vmap over an un-batched key, vec_env.py:75-77)",
                "storage": "bits (default): 1 bit per allele between steps, byte-layout observation written every step",
                "sharding": "envs across ranks, no data-path collective",
            },
            "roofline": {
                "kernel": "packed_cross_kernel<TT=1,MW=4,R=2> (row role: bit-plane parents -> select -> byte-layout "
                          "observation + bit-plane state + GEBV, 2 rows per wave; mask role: next generation's Threefry)",
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms,
                "kernel_crosses_per_s": crosses_per_step / (kern_ms * 1e-3),
            },
        }
        if ti is not None:
            line["independent_rng"] = {
                "note": "one Threefry stream per cross (env_rng='independent', an extension); VALU-bound",
                "value": world * args.steps * crosses_per_step / float(ti.item()),
                "ms_per_step": float(ti.item()) / args.steps * 1e3,
                "kernel_ms": kern_ms_i,
                "achieved_GBps": alg_bytes / (kern_ms_i * 1e-3) / 1e9,
            }
        if variants:
            variants["byte_layout_storage"]["note"] = "storage='bytes': the byte layout is the state; cross_kernel<1,4,2>"
            variants["lazy_observations"]["note"] = ("storage='bits', observations='lazy': the step hands out a handle "
                                                      "that expands to the byte layout on demand")
            line.update(variants)
        if world == 1 and not args.no_cpu_baseline and not args.headline_only:
            line["cpu_baseline"] = cpu_baseline(gm, geno, 0)
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
