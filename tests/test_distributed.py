"""CPU: the N > 1 paths under gloo with world_size 2 (one process per "GPU"; the engine is the
oracle-backed test double).  Env-sharded stepping has no collective; the population-sharded
generation uses all-gathers for top-k candidates and selected parents.

The workers take the engine as a parameter: tests/test_gpu_distributed.py runs the SAME assertions with the HIP
engine, two ranks sharing one GPU (gloo carrying the device tensors through the host: RCCL refuses two ranks per
device)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
M, N = 96, 40


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(rank, world, port):
    for p in (str(ROOT), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _engine_kw(engine_kind, gm):
    """engine=... for Simulator / the envs: the oracle-backed double, or nothing (= the HIP engine on cuda:0)."""
    if engine_kind == "hip":
        torch.cuda.set_device(0)
        return {}
    from oracle_engine import OracleEngine

    return {"engine": OracleEngine(gm.recombination_vec, gm.marker_effects)}


def _env_sharded_worker(rank, world, port, q, env_rng="shared", engine_kind="oracle", shape=None):
    _setup(rank, world, port)
    # (germplasm rows, markers, envs per rank, individuals per env, selection k, selection crosses)
    N, M, EL, n, K, NC = shape or (globals()["N"], globals()["M"], 3, 8, 4, 5)

    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
    from breedgym_amd.vector import SelectionScores
    from breedgym_amd.vector.distributed import DistributedBreedGym
    from oracle import c_oracle as co

    gm = synthetic_genetic_map(M, n_chr=2, seed=1)
    geno = synthetic_genotypes(N, M, seed=2)
    E = EL * world
    env = DistributedBreedGym(EL, geno, individual_per_gen=n, genetic_map=gm, reward_shaping=True, env_rng=env_rng,
                              **_engine_kw(engine_kind, gm))
    assert env.num_envs == E and env.local_slice() == slice(EL * rank, EL * rank + EL)
    pop, _ = env.reset(seed=10)  # the same seed on every rank
    # WORLD-SIZE INVARIANCE: what ONE VecBreedGym of 3 * world envs computes, this rank's envs of it
    mode = 0 if env_rng == "shared" else 1
    f, env_key = co.founders(co.key_from_seed(10), E, N, n)
    ref = geno.view(np.uint8)[f]
    assert np.array_equal(np.asarray(pop).view(np.uint8), ref[env.local_slice()])
    key = co.key_from_seed(10)
    arng = np.random.default_rng(0)
    for step in range(3):
        global_actions = arng.integers(0, n, (E, n, 2), dtype=np.int32)
        new_pop, r, ter, tru, infos = env.step(global_actions)  # takes its own slice
        ref, key = co.cross(ref, global_actions, gm.recombination_vec, key, mode)
        assert np.array_equal(np.asarray(new_pop).view(np.uint8), ref[env.local_slice()]), (rank, step)
        assert np.array_equal(env.env.simulator.random_key, key)
    allr = env.gather_rewards(r)
    assert allr.shape == (E,) and np.allclose(allr[env.local_slice()], r)
    assert np.allclose(allr, co.gebv(ref, gm.marker_effects).max(axis=(1, 2)))
    best = env.best_reward(r)
    assert abs(best - allr.max()) < 1e-6
    g = env.all_gather(infos["GEBV"])
    assert g.shape == (E, n, 1)
    # the selection wrapper draws its pair shuffles through the same env window
    w = SelectionScores(env.env, k=K, n_crosses=NC)
    scores = co.gebv(ref, gm.marker_effects)[..., 0]
    low = w._convert_actions(scores[env.local_slice()])
    low = low.cpu().numpy() if isinstance(low, torch.Tensor) else np.asarray(low)
    ref_low, ref_env_key = co.selection_actions(scores, K, NC, env_key)
    assert np.array_equal(low, ref_low[env.local_slice()]) and np.array_equal(env.env.random_key, ref_env_key)
    q.put((rank, allr.tolist(), best))
    dist.barrier()
    dist.destroy_process_group()


def _pop_sharded_worker(rank, world, port, q, storage="bytes", engine_kind="oracle", shape=None):
    _setup(rank, world, port)
    N, M, K = shape or (globals()["N"], globals()["M"], 9)  # (individuals, markers, winners per generation)

    from breedgym_amd.simulator import Simulator
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
    from breedgym_amd.vector.distributed import ShardedPopulation
    from oracle import c_oracle as co

    gm = synthetic_genetic_map(M, n_chr=2, seed=1)
    geno = synthetic_genotypes(N, M, seed=2).view(np.uint8)  # the whole population, known to the test only
    n_local = N // world
    mine = slice(rank * n_local, (rank + 1) * n_local)
    sim = Simulator(genetic_map=gm, seed=0, **_engine_kw(engine_kind, gm))
    rows = sim.engine.upload_population(geno[mine])
    sp = ShardedPopulation(sim, rows, storage=storage)
    scores_all = co.gebv(geno, gm.marker_effects)[:, 0]
    scores_all[7] = scores_all[31] = scores_all[3]  # ties across shards resolve to the lower global index
    scores_all[5] = scores_all[6]  # and inside a shard
    for k in (1, 5, 13, n_local, n_local + 4):
        got = sp.global_topk(torch.from_numpy(scores_all[mine].copy()), k)
        assert np.array_equal(got.cpu().numpy(), co.topk(scores_all, k)), k
    best = sp.global_topk(sp.gebv()[:, 0], K)
    parents = sp.gather_rows(best)
    assert np.array_equal(parents.cpu().numpy()[:, : 2 * M].reshape(K, M, 2), geno[best.cpu().numpy()])
    # WORLD-SIZE INVARIANCE over four generations: the same key on every rank, every rank's rows of one global
    # pair table -> the generation one device would compute for the whole population, row for row
    key = co.key_from_seed(100).copy()
    ref, refkey = geno, co.key_from_seed(100)
    for gen in range(4):
        pairs_global = np.random.default_rng(gen).integers(0, K, (N, 2), dtype=np.int32)
        if gen == 0:
            assert sp._cap_hint is None  # the first generation sizes its blocks exactly
        elif gen == 2:
            sp._cap_hint = 1  # a guess that is too small: the generation is redone with the exact size, same results
        g = sp.select_and_cross(K, torch.from_numpy(pairs_global[mine]), key)
        assert sp.redone == (1 if gen >= 2 else 0) and sp._cap_hint >= 1
        true_best = co.topk(co.gebv(ref, gm.marker_effects)[:, 0], K)
        out, refkey = co.cross(ref[true_best][None], pairs_global[None], gm.recombination_vec, refkey, 1)
        ref = out[0]
        assert np.array_equal(sp.rows.cpu().numpy()[:, : 2 * M].reshape(n_local, M, 2), ref[mine]), (rank, gen)
        assert np.array_equal(key, refkey) and np.allclose(np.asarray(g), co.gebv(ref[mine], gm.marker_effects))
    q.put((rank, best.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _run(worker, world=2, extra=(), join_s=120):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q, *extra)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(join_s)
    for p in procs:
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0, f"worker exited with {p.exitcode}"
    return sorted(q.get(timeout=5) for _ in range(world))


@pytest.mark.timeout(180)
@pytest.mark.parametrize("env_rng", ["shared", "independent"])
def test_env_sharded_two_ranks(env_rng):
    out = _run(_env_sharded_worker, extra=(env_rng,))
    assert out[0][1] == out[1][1]  # every rank sees the same gathered rewards
    assert out[0][2] == out[1][2]


@pytest.mark.timeout(180)
@pytest.mark.parametrize("storage", ["bytes", "bits"])
def test_population_sharded_two_ranks(storage):
    out = _run(_pop_sharded_worker, extra=(storage,))
    assert out[0][1] == out[1][1]  # identical global top-k on every rank


@pytest.mark.timeout(240)
def test_env_sharded_three_ranks():
    """An odd number of ranks: the env window of every rank is a third of one vector env of nine."""
    out = _run(_env_sharded_worker, world=3, extra=("shared",), join_s=180)
    assert out[0][1] == out[1][1] == out[2][1]
    assert out[0][2] == out[1][2] == out[2][2]


@pytest.mark.timeout(240)
def test_population_sharded_four_ranks():
    """Four shards (configs[3] has eight): winners from every shard, ties across shards, the speculated block size and
    its redo -- the same assertions as with two ranks, row for row against one device's generation."""
    out = _run(_pop_sharded_worker, world=4, extra=("bits",), join_s=180)
    assert out[0][1] == out[1][1] == out[2][1] == out[3][1]


@pytest.mark.timeout(300)
def test_population_sharded_eight_ranks():
    """configs[3]'s own rank count: eight shards of five rows, winners from most of them, fewer local rows than k."""
    out = _run(_pop_sharded_worker, world=8, extra=("bits",), join_s=240)
    assert all(o[1] == out[0][1] for o in out)


@pytest.mark.timeout(300)
def test_env_sharded_eight_ranks():
    """The width the driver's scaling run ends at: eight env windows of one vector env of 24."""
    out = _run(_env_sharded_worker, world=8, extra=("shared",), join_s=240)
    assert all(o[1] == out[0][1] and o[2] == out[0][2] for o in out)


@pytest.mark.timeout(300)
def test_population_sharded_two_ranks_longer_rows():
    """Rows beyond one 16-byte-granular KB (M = 4,160: bit-plane rows of 1,040 bytes), more winners than a shard has rows
    to give on average (k = 24 of 2 x 32), both storages' shared code path through the shape parameter the GPU suite
    uses for configs[3]'s row length."""
    out = _run(_pop_sharded_worker, extra=("bits", "oracle", (64, 4160, 24)), join_s=240)
    assert out[0][1] == out[1][1]
