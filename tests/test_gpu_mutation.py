"""GPU: ``Simulator(mutation=p)`` -- chromax's per-allele flip probability, which the reference forwards through
``**kwargs`` (breedgym/vector/vec_env.py:39,45) -- on every output of the cross calls (byte layout, bit planes, byte
observation next to bit planes, GEBV, reward) in all key modes, both threefry layouts, across sharding windows, and
through the envs, against the oracle (tests/test_oracle_kat.py::test_mutation_c_equals_numpy pins what the oracle
does: flips = uniform(split(gamete_key, 2)[1], (M,)) < p)."""
import numpy as np
import pytest
import torch
from oracle_engine import OracleEngine

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu


def gebv_close(got, ref64):
    return bool(np.all(np.abs(np.asarray(got, np.float64) - ref64) <= 1e-5 * np.maximum(np.abs(ref64), 1.0)))


@pytest.mark.parametrize("layout", ["original", "partitionable"])
@pytest.mark.parametrize("M", [1, 2, 7, 63, 64, 65, 129, 1000, 1001, 10000, 16392, 50001])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_cross_with_mutation_vs_oracle(M, mode, layout, rng):
    from breedgym_amd import HipEngine

    co.set_threefry_layout(layout)
    try:
        E, NP, NO = (3, 7, 9) if M <= 10000 else (2, 4, 5)
        pm = 0.5 if M < 64 else 0.01
        pool = rng.integers(0, 2, (E, NP, M, 2), dtype=np.uint8)
        if mode & 2:
            one = rng.integers(0, NP, (E, NO), dtype=np.int32)
            idx = np.stack((one, one), -1)
        else:
            idx = rng.integers(-NP, NP + 2, (E, NO, 2), dtype=np.int32)
        r = (rng.random(M) * 0.05).astype(np.float32)
        r[:: max(1, M // 5)] = 0.5
        eff = rng.laplace(0, 6, (M, 2)).astype(np.float32)
        eng = HipEngine(r, eff, threefry_layout=layout)
        eng.set_mutation(pm)
        rows = eng.upload_population(pool.reshape(E * NP, M, 2))
        packed = eng.pack(rows)
        d_idx = torch.from_numpy(idx).to(eng.device)
        key_b, key_p, ref_key = co.key_from_seed(77).copy(), co.key_from_seed(77).copy(), co.key_from_seed(77)
        n_flipped = 0
        for gen in range(3):  # the later launches read crossover masks made ahead
            out = eng.empty_population(E * NO)
            g, gp = eng.empty((E, NO, 2), torch.float32), eng.empty((E, NO, 2), torch.float32)
            rew, rewp = eng.empty_reward(E), eng.empty_reward(E)
            eng.cross(rows, E, NP, d_idx, NO, key_b, mode, out, g, rew)
            outp, ob = eng.empty_packed(E * NO), eng.empty_population(E * NO)
            eng.cross_packed(packed, E, NP, d_idx, NO, key_p, mode, outp, ob, gp, rewp)
            plain, _ = co.cross(pool, idx, r, ref_key, mode)
            ref, ref_key = co.cross(pool, idx, r, ref_key, mode, mutation=pm)
            n_flipped += int((plain != ref).sum())
            got = eng.download_population(out).view(np.uint8).reshape(ref.shape)
            assert np.array_equal(got, ref), f"M={M} mode={mode} gen={gen}"
            assert np.array_equal(eng.download_population(ob).view(np.uint8).reshape(ref.shape), ref)
            assert np.array_equal(eng.download_population(eng.unpack(outp)).view(np.uint8).reshape(ref.shape), ref)
            assert np.array_equal(key_b, ref_key) and np.array_equal(key_p, ref_key)
            g64 = co.gebv(ref, eff, want64=True)[1]
            assert gebv_close(g.cpu().numpy(), g64) and gebv_close(gp.cpu().numpy(), g64)
            rmax = g64.max(axis=(1, 2))
            assert np.allclose(rew.cpu().numpy(), rmax, rtol=1e-5) and np.allclose(rewp.cpu().numpy(), rmax, rtol=1e-5)
        assert n_flipped > 0 or M * pm < 1
        # switched off again: the plain cross
        eng.set_mutation(0.0)
        out = eng.empty_population(E * NO)
        eng.cross(rows, E, NP, d_idx, NO, key_b, mode, out, None)
        ref, _ = co.cross(pool, idx, r, ref_key, mode)
        assert np.array_equal(eng.download_population(out).view(np.uint8).reshape(ref.shape), ref)
        with pytest.raises(ValueError):
            eng.set_mutation(1.5)
    finally:
        co.set_threefry_layout("original")


def test_mutation_many_long_rows(rng):
    """The long-row launch (stream-per-lane mask role) with mutation behind it."""
    from breedgym_amd import HipEngine

    M, E, NP, NO, pm = 16523, 2, 8, 4200, 0.003
    pool = rng.integers(0, 2, (E, NP, M, 2), dtype=np.uint8)
    idx = rng.integers(0, NP, (E, NO, 2), dtype=np.int32)
    r = (rng.random(M) * 0.02).astype(np.float32)
    r[0] = 0.5
    eff = rng.laplace(0, 6, (M, 1)).astype(np.float32)
    eng = HipEngine(r, eff)
    eng.set_mutation(pm)
    packed = eng.pack(eng.upload_population(pool.reshape(E * NP, M, 2)))
    d_idx = torch.from_numpy(idx).to(eng.device)
    key, ref_key = co.key_from_seed(3).copy(), co.key_from_seed(3)
    co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    for gen in range(2):
        outp, ob, g = eng.empty_packed(E * NO), eng.empty_population(E * NO), eng.empty((E, NO, 1), torch.float32)
        eng.cross_packed(packed, E, NP, d_idx, NO, key, 0, outp, ob, g)
        ref, ref_key = co.cross(pool, idx, r, ref_key, 0, mutation=pm)
        assert np.array_equal(eng.download_population(ob).view(np.uint8).reshape(ref.shape), ref), gen
        assert np.array_equal(eng.download_population(eng.unpack(outp)).view(np.uint8).reshape(ref.shape), ref)
        assert gebv_close(g.cpu().numpy(), co.gebv(ref, eff, want64=True)[1]) and np.array_equal(key, ref_key)


def test_mutation_across_a_cross_window(rng):
    from breedgym_amd import HipEngine

    M, k_win, n_total, pm = 777, 12, 40, 0.02
    r = (rng.random(M) * 0.05).astype(np.float32)
    eff = rng.laplace(0, 6, (M, 1)).astype(np.float32)
    winners = rng.integers(0, 2, (k_win, M, 2), dtype=np.uint8)
    pairs = rng.integers(0, k_win, (n_total, 2), dtype=np.int32)
    for mode in (1, 3):
        pr = pairs if mode == 1 else np.repeat(pairs[:, :1], 2, axis=1)
        ref, kref = co.cross(winners[None], pr[None], r, co.key_from_seed(8), mode, mutation=pm)
        for lo, hi in ((0, 16), (16, 40)):
            eng = HipEngine(r, eff)
            eng.set_mutation(pm)
            eng.set_cross_window(lo, n_total)
            wp = eng.pack(eng.upload_population(winners))
            d_idx = torch.from_numpy(pr[lo:hi].copy()).to(eng.device).reshape(1, -1, 2)
            key = co.key_from_seed(8).copy()
            outp = eng.empty_packed(hi - lo)
            eng.cross_packed(wp, 1, k_win, d_idx, hi - lo, key, mode, outp, None, None)
            assert np.array_equal(eng.download_population(eng.unpack(outp)).view(np.uint8), ref[0][lo:hi]), (mode, lo)
            assert np.array_equal(key, kref)


@pytest.mark.parametrize("storage", ["bits", "bytes"])
def test_vec_env_with_mutation_equals_oracle_chain(storage):
    """VecBreedGym(mutation=p) -- the keyword travels to the simulator as in the reference -- over an episode and
    its autoreset, against the oracle chain and against the same env on the oracle-backed engine."""
    from breedgym_amd import VecBreedGym
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    N, M, E, n, pm = 40, 1300, 3, 16, 0.004
    gm = synthetic_genetic_map(M, n_chr=4, seed=3)
    geno = synthetic_genotypes(N, M, seed=4)
    kw = dict(num_envs=E, initial_population=geno, individual_per_gen=n, genetic_map=gm, num_generations=3,
              reward_shaping=True, mutation=pm)
    dev = VecBreedGym(storage=storage, **kw)
    ora = VecBreedGym(engine=OracleEngine(gm.recombination_vec, gm.marker_effects), **kw)
    assert dev.simulator.mutation == pm and dev.simulator.engine.mutation == pm
    pop, _ = dev.reset(seed=7)
    ora.reset(seed=7)
    f, env_key = co.founders(co.key_from_seed(7), E, N, n)
    ref = geno.view(np.uint8)[f]
    sim_key = co.key_from_seed(7)
    arng = np.random.default_rng(1)
    for step in range(5):
        a = arng.integers(0, n, (E, n, 2), dtype=np.int32)
        pop, rews, _, tru, infos = dev.step(a)
        po, ro, *_ = ora.step(a)
        ref, sim_key = co.cross(ref, a, gm.recombination_vec, sim_key, 0, mutation=pm)
        assert np.allclose(rews, co.gebv(ref, gm.marker_effects).max(axis=(1, 2)), rtol=1e-5) and np.allclose(rews, ro, rtol=1e-5)
        if step == 2:
            assert tru.all()
            f, env_key = co.founders(env_key, E, N, n)
            ref = geno.view(np.uint8)[f]
        assert np.array_equal(np.asarray(pop).view(np.uint8), ref), step
        assert np.array_equal(np.asarray(po).view(np.uint8), ref)
        assert np.array_equal(dev.simulator.random_key, sim_key)
