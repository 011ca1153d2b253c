"""GPU: JAX's partitionable threefry layout (the default of jax >= 0.5: ``jax_threefry_partitionable = True``)
on every device path that draws random bits -- crossover masks in all key modes and both storages, the
founder draw, ``SelectionScores``' pair shuffle, and the env end to end -- against the oracle switched to the
same layout (pinned on JAX's documented values in tests/test_oracle_kat.py)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def partitionable_oracle():
    co.set_threefry_layout("partitionable")
    yield
    co.set_threefry_layout("original")


def engine(r, eff):
    from breedgym_amd import HipEngine

    return HipEngine(r, eff, threefry_layout="partitionable")


@pytest.mark.parametrize("M", [1, 2, 63, 64, 65, 129, 1000, 1001, 10000, 16392, 50001])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_cross_partitionable_vs_oracle(M, mode, rng):
    E, NP, NO = (3, 7, 9) if M <= 10000 else (2, 4, 5)
    pool = rng.integers(0, 2, (E, NP, M, 2), dtype=np.uint8)
    if mode & 2:  # double haploid: one parent per row
        one = rng.integers(0, NP, (E, NO), dtype=np.int32)
        idx = np.stack((one, one), -1)
    else:
        idx = rng.integers(-NP, NP + 2, (E, NO, 2), dtype=np.int32)
    r = (rng.random(M) * 0.05).astype(np.float32)
    r[:: max(1, M // 5)] = 0.5
    r[rng.random(M) < 0.16] = 0.0
    if M > 70:
        r[17], r[M - 3] = 1.0, 1.5  # always-recombining markers go through the scalar site words
    eff = rng.laplace(0, 6, (M, 1)).astype(np.float32)
    eng = engine(r, eff)
    rows = eng.upload_population(pool.reshape(E * NP, M, 2))
    packed = eng.pack(rows)
    d_idx = torch.from_numpy(idx).to(eng.device)
    for prefetch in (True, False):
        eng.set_mask_prefetch(prefetch)
        key_b = co.key_from_seed(77).copy()
        key_p = key_b.copy()
        ref_key = co.key_from_seed(77)
        for gen in range(3):  # the second and third launch read masks made ahead (when prefetch is on)
            out = eng.empty_population(E * NO)
            g = eng.empty((E, NO, 1), torch.float32)
            eng.cross(rows, E, NP, d_idx, NO, key_b, mode, out, g)
            outp, ob = eng.empty_packed(E * NO), eng.empty_population(E * NO)
            eng.cross_packed(packed, E, NP, d_idx, NO, key_p, mode, outp, ob, g)
            ref, ref_key = co.cross(pool, idx, r, ref_key, mode)
            got = eng.download_population(out).view(np.uint8).reshape(ref.shape)
            assert np.array_equal(got, ref), f"M={M} mode={mode} gen={gen} prefetch={prefetch}"
            assert np.array_equal(eng.download_population(ob).view(np.uint8).reshape(ref.shape), ref)
            assert np.array_equal(eng.download_population(eng.unpack(outp)).view(np.uint8).reshape(ref.shape), ref)
            assert np.array_equal(key_b, ref_key) and np.array_equal(key_p, ref_key)
    # the other layout is a different stream
    if M >= 64:
        eng.set_threefry_layout("original")
        out = eng.empty_population(E * NO)
        k = co.key_from_seed(77).copy()
        eng.cross(rows, E, NP, d_idx, NO, k, mode, out, None)
        ref, _ = co.cross(pool, idx, r, co.key_from_seed(77), mode)
        assert not np.array_equal(eng.download_population(out).view(np.uint8).reshape(ref.shape), ref)
        co.set_threefry_layout("original")
        ref0, k0 = co.cross(pool, idx, r, co.key_from_seed(77), mode)
        assert np.array_equal(eng.download_population(out).view(np.uint8).reshape(ref.shape), ref0) and np.array_equal(k, k0)


@pytest.mark.parametrize("N,n,E", [(1, 1, 1), (2, 2, 3), (370, 200, 4), (371, 371, 5), (1625, 64, 2), (8192, 100, 2),
                                   (9000, 50, 2)])
def test_founders_partitionable(N, n, E):
    eng = engine(np.full(8, 0.1, np.float32), np.ones((8, 1), np.float32))
    got, key = eng.founders(co.key_from_seed(5), E, N, n)
    ref, refkey = co.founders(co.key_from_seed(5), E, N, n)
    assert np.array_equal(got.cpu().numpy().reshape(E, n), ref) and np.array_equal(key, refkey)


@pytest.mark.parametrize("E,n,k,nc", [(1, 2, 2, 1), (3, 50, 5, 7), (4, 210, 10, 20), (2, 1000, 40, 700)])
def test_selection_actions_partitionable(E, n, k, nc, rng):
    eng = engine(np.full(8, 0.1, np.float32), np.ones((8, 1), np.float32))
    scores = rng.standard_normal((E, n)).astype(np.float32)
    got, key = eng.selection_actions(torch.from_numpy(scores).to(eng.device), k, nc, co.key_from_seed(9))
    ref, refkey = co.selection_actions(scores, k, nc, co.key_from_seed(9))
    assert np.array_equal(got.cpu().numpy(), ref) and np.array_equal(key, refkey)


def test_env_partitionable_equals_oracle_chain():
    """VecBreedGym(threefry_layout="partitionable") with the SelectionScores policy of reference
    tests/test_vec.py:136-154, over an episode and its autoreset."""
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
    from breedgym_amd.vector import SelectionScores

    N, M, E, n = 60, 1300, 3, 24
    gm = synthetic_genetic_map(M, n_chr=4, seed=3)
    geno = synthetic_genotypes(N, M, seed=4)
    env = SelectionScores(k=5, n_crosses=8, num_envs=E, initial_population=geno, individual_per_gen=n, genetic_map=gm,
                          num_generations=3, reward_shaping=True, threefry_layout="partitionable")
    assert env.simulator.engine.threefry_layout == "partitionable"
    pop, infos = env.reset(seed=7)
    f, env_key = co.founders(co.key_from_seed(7), E, N, n)
    ref = geno.view(np.uint8)[f]
    assert np.array_equal(np.asarray(pop).view(np.uint8), ref)
    sim_key = co.key_from_seed(7)
    for step in range(5):
        low, env_key = co.selection_actions(co.gebv(ref, gm.marker_effects)[..., 0], 5, 8, env_key)
        ref, sim_key = co.cross(ref, low, gm.recombination_vec, sim_key, 0)
        pop, rews, _, tru, infos = env.step(infos["GEBV"][..., 0])
        assert np.allclose(rews, co.gebv(ref, gm.marker_effects).max(axis=(1, 2)), rtol=1e-5)
        if step == 2:
            assert tru.all()
            f, env_key = co.founders(env_key, E, N, n)
            ref = geno.view(np.uint8)[f]
            infos = env.reset_infos
        assert np.array_equal(np.asarray(pop).view(np.uint8), ref), step
        assert np.array_equal(env.simulator.random_key, sim_key) and np.array_equal(env.unwrapped.random_key, env_key)
