"""GPU: the BASELINE.json configurations at their real shapes on the path the product uses, and the
single-env / wrapper surface on the device, against the oracle.

* configs[1] (c2): the headline env -- 16 envs x 256 individuals of a 370 x 10k germplasm, all defaults
  (shared keys, bit-plane state, byte observation) -- stepped through a whole episode and over the
  autoreset, byte for byte equal to the oracle chain ``founders -> cross -> cross ...``
  (the reference's tests/test_vec.py:115-133 loop with the oracle in place of the chromax constants);
* configs[2] (c3): 2048 x 100k germplasm, 4 envs x 2048 = 8192 crosses, default path, two generations
  (the second one reads run-ahead masks), every cross against the oracle;
* configs[3] (c4): one rank's generation of the 8-way population-sharded flow at 8192 x 100k, k = 6553;
* configs[4] (c5): GEBV of 2048 x 100k x 8 traits against the f64 oracle, byte rows and bit planes;
* long rows with same-sign effects (the fixed-point recombination must not wrap);
* ``BreedGym``, ``SimplifiedBreedGym``, ``KBestBreedGym``, ``PairScores``, ``WheatBreedGym`` on the
  device against the same classes driven by the oracle-backed engine (tests/oracle_engine.py):
  reference tests/test_env.py:103-119, tests/test_wrappers.py:7-88, tests/test_vec.py:186-204.
"""
import numpy as np
import pytest
import torch
from oracle_engine import OracleEngine

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu

GEBV_RTOL = 1e-5  # BASELINE.json north_star: "within 1e-5 on GEBV"


def gebv_close(got, ref64):
    scale = np.maximum(np.abs(ref64), 1.0)
    return bool(np.all(np.abs(np.asarray(got, np.float64) - ref64) <= GEBV_RTOL * scale))


def u8(x):
    return np.asarray(x).view(np.uint8)


# ---- configs[1]: the headline env ---------------------------------------------------------------------


def test_c2_headline_env_equals_oracle_chain():
    from breedgym_amd import VecBreedGym
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    N, M, E, n, G = 370, 10_000, 16, 256, 10
    gm = synthetic_genetic_map(M, n_chr=10, n_traits=1, seed=1)
    geno = synthetic_genotypes(N, M, seed=0)
    env = VecBreedGym(num_envs=E, initial_population=geno, individual_per_gen=n, num_generations=G,
                      reward_shaping=True, genetic_map=gm)  # every other keyword at its default
    assert env.storage == "bits" and env.observations == "eager"
    pop, info = env.reset(seed=100)
    f, env_key = co.founders(co.key_from_seed(100), E, N, n)
    ref = geno.view(np.uint8)[f]
    assert np.array_equal(u8(pop), ref)
    assert gebv_close(np.asarray(info["GEBV"]), co.gebv(ref, gm.marker_effects, want64=True)[1])
    sim_key = co.key_from_seed(100)
    arng = np.random.default_rng(2)
    for step in range(G + 3):  # a whole episode, the autoreset, and three steps of the next one
        a = arng.integers(0, n, (E, n, 2), dtype=np.int32)
        pop, rews, ter, tru, info = env.step(a)
        ref, sim_key = co.cross(ref, a, gm.recombination_vec, sim_key, 0)
        g64 = co.gebv(ref, gm.marker_effects, want64=True)[1]
        assert gebv_close(np.asarray(info["GEBV"]), g64), f"step {step}"
        assert np.allclose(rews, g64.max(axis=(1, 2)), rtol=GEBV_RTOL)
        assert np.array_equal(env.simulator.random_key, sim_key)
        if step == G - 1:  # horizon: the observation is the next episode's founders (vec_env.py:102-107)
            assert tru.all() and env.step_idx == 0
            f, env_key = co.founders(env_key, E, N, n)
            ref = geno.view(np.uint8)[f]
        else:
            assert not tru.any()
        assert np.array_equal(u8(pop), ref), f"step {step}: offspring differ from the oracle chain"
        assert np.array_equal(env.random_key, env_key)
        # the bit-plane state behind the observation says the same
        assert np.array_equal(u8(env.simulator.engine.as_genotypes(env.simulator.engine.unpack(env._prows), E, n)), ref)


# ---- configs[2]: wheat scale on the default path ------------------------------------------------------


@pytest.mark.parametrize("observations", ["eager", "lazy"])
def test_c3_default_path_full_shape(observations):
    from breedgym_amd import VecBreedGym
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    N, M, E, n = 2048, 100_000, 4, 2048
    gm = synthetic_genetic_map(M, n_chr=21, n_traits=1, seed=1)
    geno = synthetic_genotypes(N, M, seed=0)
    env = VecBreedGym(num_envs=E, initial_population=geno, individual_per_gen=n, num_generations=10,
                      reward_shaping=True, genetic_map=gm, observations=observations)
    co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    pop, info = env.reset(seed=3)
    f, _ = co.founders(co.key_from_seed(3), E, N, n)
    ref = geno.view(np.uint8)[f]
    del geno
    assert np.array_equal(u8(pop), ref)
    assert gebv_close(np.asarray(info["GEBV"]), co.gebv(ref, gm.marker_effects, want64=True)[1])
    key = co.key_from_seed(3)
    arng = np.random.default_rng(5)
    for gen in range(2):  # generation 1 reads the masks generation 0's launch made ahead
        a = arng.integers(0, n, (E, n, 2), dtype=np.int32)
        pop, rews, _, _, info = env.step(a)
        ref, key = co.cross(ref, a, gm.recombination_vec, key, 0)
        assert np.array_equal(env.simulator.random_key, key)
        got = u8(pop)
        assert got.shape == ref.shape
        for e in range(E):  # env by env: keeps the temporaries of the comparison small
            assert np.array_equal(got[e], ref[e]), f"generation {gen}, env {e}: offspring differ from the oracle"
        del got
        g64 = co.gebv(ref, gm.marker_effects, want64=True)[1]
        assert gebv_close(np.asarray(info["GEBV"]), g64)
        assert np.allclose(rews, g64.max(axis=(1, 2)), rtol=GEBV_RTOL)


# ---- configs[4]: eight traits at 100k markers ----------------------------------------------------------


def test_c5_multi_trait_gebv_full_shape():
    from breedgym_amd import HipEngine
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    N, M, T = 2048, 100_000, 8
    gm = synthetic_genetic_map(M, n_chr=21, n_traits=T, seed=1)
    geno = synthetic_genotypes(N, M, seed=0).view(np.uint8)
    eng = HipEngine(gm.recombination_vec, gm.marker_effects)
    rows = eng.upload_population(geno)
    co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    g64 = co.gebv(geno, gm.marker_effects, want64=True)[1]
    assert gebv_close(eng.gebv(rows).cpu().numpy(), g64)  # bh_gebv, byte-layout rows
    packed = eng.pack(rows)
    g = eng.empty((N, T), torch.float32)
    eng.gather_packed(packed, None, N, None, None, g)  # GEBV straight from the bit planes
    assert gebv_close(g.cpu().numpy(), g64)
    # and through an index (the reset gather), a ragged row count
    idx = torch.from_numpy(np.random.default_rng(0).integers(0, N, 777, dtype=np.int32)).to(eng.device)
    g2 = eng.empty((777, T), torch.float32)
    outp = eng.empty_packed(777)
    eng.gather_packed(packed, idx, 777, outp, None, g2)
    assert gebv_close(g2.cpu().numpy(), g64[idx.cpu().numpy()])
    assert torch.equal(outp, packed[idx.long()])


@pytest.mark.parametrize("T", [1, 3])
@pytest.mark.parametrize("rows,M", [(1024, 32_768), (1024, 32_767), (2050, 16_385), (2047, 16_385), (130, 262_144)])
def test_byte_row_gebv_on_both_sides_of_the_bit_plane_route(rows, M, T, rng):
    """bh_gebv on byte-layout rows: populations of 2^25 row x markers and more are packed to bit planes and take the
    matrix-core pass over those, smaller ones the byte-row kernels -- the same GEBV on both sides of the switch."""
    from breedgym_amd import HipEngine

    r = np.full(M, 0.01, np.float32)
    eff = rng.laplace(0, 6, (M, T)).astype(np.float32)
    geno = rng.integers(0, 2, (rows, M, 2), dtype=np.uint8)
    eng = HipEngine(r, eff)
    co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    g64 = co.gebv(geno, eff, want64=True)[1]
    d = eng.upload_population(geno)
    assert gebv_close(eng.gebv(d).cpu().numpy(), g64)
    assert gebv_close(eng.gebv(d).cpu().numpy(), g64)  # again: the scratch bit planes are reused
    # a smaller population through the same engine afterwards (scratch larger than needed)
    assert gebv_close(eng.gebv(d[: rows // 2].contiguous()).cpu().numpy(), g64[: rows // 2])


@pytest.mark.parametrize("T", [1, 2, 3, 8])
@pytest.mark.parametrize("M", [131_072, 200_001])
def test_same_sign_effects_on_long_rows_do_not_wrap(T, M):
    """Every fixed-point route recombines digit sums in int64: sum_m dosage * round(e * 2^s) must stay
    below 2^63 for ANY genotype.  All-ones genotypes x constant positive effects is the extreme row."""
    from breedgym_amd import HipEngine

    rows_n = 70
    eff = np.full((M, T), 0.9, np.float32)
    eff[:, T - 1] = -475.0  # and the extreme negative row, at the heavy-tail magnitude of the Yield column
    r = np.full(M, 1e-4, np.float32)
    eng = HipEngine(r, eff)
    pop = np.ones((rows_n, M, 2), np.uint8)
    pop[1::2, ::3, 1] = 0
    rows = eng.upload_population(pop)
    g64 = co.gebv(pop, eff, want64=True)[1]
    assert np.abs(g64).max() > 2e5
    assert gebv_close(eng.gebv(rows).cpu().numpy(), g64)
    packed = eng.pack(rows)
    g = eng.empty((rows_n, T), torch.float32)
    outp = eng.empty_packed(rows_n)
    eng.gather_packed(packed, None, rows_n, outp, None, g)  # the pipelined stream (digit products) for T <= 2
    assert gebv_close(g.cpu().numpy(), g64)
    # fused into a cross, with and without the byte observation
    idx = torch.from_numpy(np.stack((np.arange(rows_n), np.arange(rows_n)), 1).astype(np.int32)[None]).to(eng.device)
    for want_bytes in (True, False):
        for rep in range(2):  # the second call takes the two-role launch's run-ahead masks
            key = co.key_from_seed(1 + rep).copy() if rep == 0 else key
            gc = eng.empty((rows_n, T), torch.float32)
            ob = eng.empty_population(rows_n) if want_bytes else None
            eng.cross_packed(packed, 1, rows_n, idx, rows_n, key, 0, outp, ob, gc)
            # selfing an individual: homozygous sites stay, so GEBV of the offspring of all-ones rows is the row's
            off = eng.download_population(eng.unpack(outp)).view(np.uint8)
            assert gebv_close(gc.cpu().numpy(), co.gebv(off, eff, want64=True)[1])
    out = eng.empty_population(rows_n)
    gb = eng.empty((rows_n, T), torch.float32)
    key = co.key_from_seed(9).copy()
    for rep in range(2):
        eng.cross(rows, 1, rows_n, idx, rows_n, key, 0, out, gb)
        assert gebv_close(gb.cpu().numpy(), co.gebv(eng.download_population(out).view(np.uint8), eff, want64=True)[1])


# ---- the single-env surface and the wrappers on the device, against the oracle-backed twin ------------


M1, N1 = 1000, 60


@pytest.fixture(scope="module")
def small():
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    gm = synthetic_genetic_map(M1, n_chr=4, n_traits=1, seed=5)
    geno = synthetic_genotypes(N1, M1, seed=6)
    return gm, geno


def twins(cls, small, gm=None, **kw):
    """The same environment twice: on the device, and on the CPU oracle through the engine seam."""
    gm0, geno = small
    gm = gm or gm0
    dev = cls(initial_population=geno, genetic_map=gm, **kw)
    ora = cls(initial_population=geno, genetic_map=gm, engine=OracleEngine(gm.recombination_vec, gm.marker_effects), **kw)
    assert dev.simulator.engine.device.type == "cuda" and type(dev.simulator.engine).__name__ == "HipEngine"
    return dev, ora


def test_breedgym_on_device_equals_oracle(small):  # reference tests/test_env.py:22-34, 103-119
    from breedgym_amd import BreedGym

    gm, geno = small
    dev, ora = twins(BreedGym, small, reward_shaping=False)
    for env in (dev, ora):
        env.reset(seed=7)
    action = np.array([[1, 2], [1, 5], [1, 7], [2, 5], [2, 9], [4, 7], [4, 8], [5, 9], [6, 8], [6, 9]])
    ref, key = geno.view(np.uint8)[None], co.key_from_seed(7)
    for i in range(10):
        pd_, rd, terd, trud, infd = dev.step(action)
        po, ro, _, truo, info = ora.step(action)
        ref, key = co.cross(ref, action[None].astype(np.int32), gm.recombination_vec, key, 0)
        assert np.array_equal(u8(pd_), ref[0]) and np.array_equal(u8(po), ref[0])
        assert trud == truo == (i == 9) and not terd
        assert gebv_close(infd["GEBV"].to_numpy(), co.gebv(ref[0], gm.marker_effects, want64=True)[1])
        assert list(infd["GEBV"].columns) == gm.trait_names
        assert abs(rd - ro) <= 1e-5 * max(1.0, abs(ro)) and ((rd == 0) == (i < 9))
        assert np.array_equal(dev.simulator.random_key, key)
    assert abs(rd - co.gebv(ref[0], gm.marker_effects, want64=True)[1].mean()) < 1e-5 * max(1.0, abs(rd))
    # the population changes size with the action (test_num_progenies), corrcoef is cached and bounded
    for n in (1, 5, 33):
        dev.reset(seed=1)
        ora.reset(seed=1)
        a = np.random.default_rng(n).integers(0, N1, (n, 2))
        pd_, *_ = dev.step(a)
        po, *_ = ora.step(a)
        assert pd_.shape == (n, M1, 2) and np.array_equal(u8(pd_), u8(po))
        assert np.allclose(np.asarray(dev.corrcoef), np.asarray(ora.corrcoef), atol=1e-5)
        assert dev.corrcoef is dev.corrcoef and dev.GEBV is dev.GEBV
    # a random founder subset (reset options), then a step
    sd, _ = dev.reset(seed=4, options={"n_individuals": 9})
    so, _ = ora.reset(seed=4, options={"n_individuals": 9})
    assert sd.shape == (9, M1, 2) and np.array_equal(u8(sd), u8(so))
    assert gebv_close(dev.GEBV.to_numpy(), co.gebv(u8(so), gm.marker_effects, want64=True)[1])


@pytest.mark.parametrize("which", ["simplified", "kbest"])
def test_simplified_and_kbest_on_device_equal_oracle(small, which):  # reference tests/test_wrappers.py:7-88
    from breedgym_amd import BreedGym
    from breedgym_amd.wrappers import KBestBreedGym, SimplifiedBreedGym

    gm, geno = small
    n = 30
    bd, bo = twins(BreedGym, small)
    W = SimplifiedBreedGym if which == "simplified" else KBestBreedGym
    dev, ora = W(bd, individual_per_gen=n), W(bo, individual_per_gen=n)
    od, _ = dev.reset(seed=7)
    oo, _ = ora.reset(seed=7)
    actions = ([{"n_bests": 10, "n_crosses": 20}, {"n_bests": 8, "n_crosses": 28}, {"n_bests": 2, "n_crosses": 1}]
               if which == "simplified" else [5, 2, 8, 6])
    for a in [None] + actions:
        if a is not None:
            od, rd, _, td, _ = dev.step(a)
            oo, ro, _, to, _ = ora.step(a)
            assert td == to and abs(rd - ro) <= 1e-5 * max(1.0, abs(ro))
        assert len(od["GEBV"]) == n and len(od["corrcoef"]) == n
        assert np.array_equal(u8(dev.unwrapped.population), u8(ora.unwrapped.population))
        g64 = co.gebv(u8(ora.unwrapped.population), gm.marker_effects, want64=True)[1][:, 0]
        assert gebv_close(od["GEBV"], g64)
        c = np.asarray(od["corrcoef"])
        assert np.allclose(c, np.asarray(oo["corrcoef"]), atol=1e-5) and np.all(np.abs(c) <= 1 + 1e-6)
    for bad in (({"n_bests": 10, "n_crosses": 31}, {"n_bests": 2, "n_crosses": 10}, {"n_bests": 1, "n_crosses": 1},
                 {"n_bests": 500, "n_crosses": 10}) if which == "simplified" else (1, 9)):
        with pytest.raises(Exception):
            dev.step(bad)


def _pair_actions_restated(scores, n):
    """vec_wrappers.py:100-112 in NumPy: top-n of the flattened pair scores, ceil(softmax * n) offspring each,
    repeat, cut at n."""
    out = np.empty((scores.shape[0], n, 2), np.int32)
    for e in range(scores.shape[0]):
        flat = scores[e].reshape(-1).astype(np.float32)
        best = co.topk(flat, n)
        v = flat[best]
        ex = np.exp(v - v.max())
        sm = ex / ex.sum(dtype=np.float64).astype(np.float32)
        rep = np.ceil(sm * np.float32(n)).astype(np.int64)
        pairs = np.stack((best // n, best % n), 1)  # (flat index // n, flat index % n), vec_wrappers.py:104-105
        out[e] = np.repeat(pairs, rep, axis=0)[:n]
    return out


def test_pair_scores_on_device_equals_oracle(small):  # reference tests/test_vec.py:186-204
    from breedgym_amd.vector import PairScores

    gm, geno = small
    E, n = 3, 24
    dev, ora = twins(PairScores, small, num_envs=E, individual_per_gen=n, num_generations=4, reward_shaping=True)
    _, infd = dev.reset(seed=7)
    _, info = ora.reset(seed=7)
    for step in range(4):
        g = np.asarray(info["GEBV"])[..., 0]
        scores = g[:, :, None] + g[:, None, :]  # np.add.outer per env, as in the reference's test
        od, rd, _, td, infd = dev.step(scores)
        oo, ro, _, to, info = ora.step(scores)
        low = np.asarray(infd["low_level_actions"])
        assert np.array_equal(low, _pair_actions_restated(scores, n)), f"step {step}"
        assert np.array_equal(low, np.asarray(info["low_level_actions"]))
        assert np.array_equal(u8(od), u8(oo)) and np.array_equal(td, to)
        assert np.allclose(rd, ro, rtol=1e-5)
        assert gebv_close(np.asarray(infd["GEBV"]), np.asarray(info["GEBV"], np.float64))


def test_wheat_breed_gym_on_device_equals_oracle(small):  # breeding_programs_env.py:39-72
    from breedgym_amd import VecBreedGym
    from breedgym_amd.vector.breeding_programs_env import WheatBreedGym

    gm, geno = small
    E, n = 2, 12
    bd, bo = twins(VecBreedGym, small, num_envs=E, individual_per_gen=n, num_generations=3, reward_shaping=True)
    dev = WheatBreedGym(bd, n_lines=8, plant_per_line=6, k_per_line=2)
    ora = WheatBreedGym(bo, n_lines=8, plant_per_line=6, k_per_line=2)
    pd_, infd = dev.reset(seed=4)
    po, info = ora.reset(seed=4)
    assert np.array_equal(u8(pd_), u8(po))
    srng = np.random.default_rng(0)
    for step in range(4):  # over the horizon and the autoreset
        scores = srng.standard_normal((E, 8, 8)).astype(np.float32)  # (E, n_lines, n_lines)
        pd_, rd, _, td, infd = dev.step(scores)
        po, ro, _, to, info = ora.step(scores)
        assert np.array_equal(u8(pd_), u8(po)), f"step {step}"
        assert np.array_equal(td, to) and np.allclose(rd, ro, rtol=1e-5)
        assert gebv_close(np.asarray(infd["GEBV"]), np.asarray(info["GEBV"], np.float64))
        assert np.array_equal(bd.simulator.random_key, bo.simulator.random_key)
        assert np.array_equal(bd.random_key, bo.random_key)
        p = u8(pd_)
        if not td.all():
            assert np.array_equal(p[..., 0], p[..., 1])  # every kept plant is a doubled haploid


def test_wheat_breed_gym_bit_planes_equal_byte_rows_on_device(small):
    """the wheat generation on bit planes (the default) against the byte-layout route, both on the device, at a size
    where the doubled haploids' GEBV takes the matrix-core pass (T = 3) as well as the fused one (T = 1)"""
    from breedgym_amd import VecBreedGym
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
    from breedgym_amd.vector.breeding_programs_env import WheatBreedGym

    for T in (1, 3):
        gm = synthetic_genetic_map(2000, n_chr=5, n_traits=T, seed=3)
        geno = synthetic_genotypes(50, 2000, seed=4)
        kw = dict(num_envs=3, initial_population=geno, individual_per_gen=20, genetic_map=gm, num_generations=3,
                  reward_shaping=True)
        wb = WheatBreedGym(VecBreedGym(storage="bits", **kw), n_lines=12, plant_per_line=30, k_per_line=3)
        wy = WheatBreedGym(VecBreedGym(storage="bytes", **kw), n_lines=12, plant_per_line=30, k_per_line=3)
        wb.reset(seed=6)
        wy.reset(seed=6)
        srng = np.random.default_rng(1)
        for step in range(4):
            scores = srng.standard_normal((3, 12, 12)).astype(np.float32)
            pb, rb, _, tb, ib = wb.step(scores)
            py, ry, _, ty, iy = wy.step(scores)
            assert np.array_equal(u8(pb), u8(py)), (T, step)
            assert gebv_close(np.asarray(ib["GEBV"]), np.asarray(iy["GEBV"], np.float64)) and np.allclose(rb, ry, rtol=1e-5)
            assert np.array_equal(wb.simulator.random_key, wy.simulator.random_key)


# ---- sharding windows: N shards compute what one device would -----------------------------


@pytest.mark.parametrize("layout", ["original", "partitionable"])
def test_sharding_windows_on_device(layout, rng):
    """Two shards played one after the other on one GPU, each through its window: founder draws and selection
    shuffles (env window), independent-key crosses of env shards (env window) and of one row-sharded batch (cross
    window, byte layout and bit planes, over generations so that masks made ahead are used) -- all equal to the
    oracle's result for the WHOLE, cut at the shard boundary."""
    from breedgym_amd import HipEngine

    co.set_threefry_layout(layout)
    try:
        M, N, E, n = 777, 50, 6, 10
        r = (rng.random(M) * 0.05).astype(np.float32)
        r[0] = 0.5
        eff = rng.laplace(0, 6, (M, 1)).astype(np.float32)
        geno = rng.integers(0, 2, (N, M, 2), dtype=np.uint8)
        f_ref, key_ref = co.founders(co.key_from_seed(3), E, N, n)
        scores = rng.standard_normal((E, n)).astype(np.float32)
        a_ref, akey_ref = co.selection_actions(scores, 4, 5, co.key_from_seed(4))
        pool = geno[f_ref]  # (E, n, M, 2)
        idx = rng.integers(0, n, (E, n, 2), dtype=np.int32)
        x_ref, xkey_ref = [], co.key_from_seed(6)
        cur = pool
        for gen in range(3):
            cur, xkey_ref = co.cross(cur, idx, r, xkey_ref, 1)
            x_ref.append(cur)
        for shard, (lo, hi) in enumerate(((0, 2), (2, 6))):  # uneven shards
            eng = HipEngine(r, eff, threefry_layout=layout)
            eng.set_env_window(lo, E)
            f, k = eng.founders(co.key_from_seed(3), hi - lo, N, n)
            assert np.array_equal(f.cpu().numpy().reshape(hi - lo, n), f_ref[lo:hi]) and np.array_equal(k, key_ref)
            a, k = eng.selection_actions(torch.from_numpy(scores[lo:hi]).to(eng.device), 4, 5, co.key_from_seed(4))
            assert np.array_equal(a.cpu().numpy(), a_ref[lo:hi]) and np.array_equal(k, akey_ref)
            rows = eng.upload_population(pool[lo:hi].reshape(-1, M, 2))
            d_idx = torch.from_numpy(idx[lo:hi].copy()).to(eng.device)
            key = co.key_from_seed(6).copy()
            for gen in range(3):
                out = eng.empty_population((hi - lo) * n)
                eng.cross(rows, hi - lo, n, d_idx, n, key, 1, out, None)
                got = eng.download_population(out).view(np.uint8).reshape(hi - lo, n, M, 2)
                assert np.array_equal(got, x_ref[gen][lo:hi]), (shard, gen)
                rows = out
            assert np.array_equal(key, xkey_ref)
        # one batch of crosses cut row-wise (population-sharded), bit planes, double haploid as well
        k_win, n_total = 12, 40
        winners = geno[:k_win]
        pairs = rng.integers(0, k_win, (n_total, 2), dtype=np.int32)
        for mode in (1, 3):
            pr = pairs if mode == 1 else np.repeat(pairs[:, :1], 2, axis=1)
            refs, kref = [], co.key_from_seed(8)
            for gen in range(2):
                o, kref = co.cross(winners[None], pr[None], r, kref, mode)
                refs.append(o[0])
            for lo, hi in ((0, 16), (16, 40)):
                eng = HipEngine(r, eff, threefry_layout=layout)
                eng.set_cross_window(lo, n_total)
                wrows = eng.upload_population(winners)
                wp = eng.pack(wrows)
                d_idx = torch.from_numpy(pr[lo:hi].copy()).to(eng.device).reshape(1, -1, 2)
                kb, kp = co.key_from_seed(8).copy(), co.key_from_seed(8).copy()
                for gen in range(2):
                    out, outp = eng.empty_population(hi - lo), eng.empty_packed(hi - lo)
                    g = eng.empty((hi - lo, 1), torch.float32)
                    eng.cross(wrows, 1, k_win, d_idx, hi - lo, kb, mode, out, g)
                    eng.cross_packed(wp, 1, k_win, d_idx, hi - lo, kp, mode, outp, None, g)
                    assert np.array_equal(eng.download_population(out).view(np.uint8), refs[gen][lo:hi]), (mode, lo, gen)
                    assert np.array_equal(eng.download_population(eng.unpack(outp)).view(np.uint8), refs[gen][lo:hi])
                    assert gebv_close(g.cpu().numpy(), co.gebv(refs[gen][lo:hi], eff, want64=True)[1])
                assert np.array_equal(kb, kref) and np.array_equal(kp, kref)
        with pytest.raises(ValueError):
            eng.cross(wrows, 1, k_win, torch.zeros((1, 30, 2), dtype=torch.int32, device=eng.device), 30, kb, 1,
                      eng.empty_population(30), None)  # 30 crosses at offset 16 exceed the window of 40
    finally:
        co.set_threefry_layout("original")


# ---- configs[3]: one rank's generation of the population-sharded flow at its real shape ---------------------


def test_c4_one_ranks_generation_full_shape():
    """BASELINE configs[3] as rank 5 of 8 sees it: an 8192 x 100k bit-plane shard of a 65,536-row population,
    GEBV of the shard, local top-6553 (with exact ties: the shard repeats 1024 distinct rows), a 6553-row winners'
    pool, and this rank's 8192 crosses drawn through its cross window of one split(key, 2 x 65,536 + 1) -- two
    generations (the second reads masks made ahead).  Sampled crosses against the oracle's meiosis for exactly
    those keys; the 8-GPU collectives themselves are covered at small shapes by tests/test_distributed.py."""
    from breedgym_amd import HipEngine
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    n_total, n_local, M, k, rank = 65_536, 8192, 100_000, 6553, 5
    gm = synthetic_genetic_map(M, n_chr=21, n_traits=1, seed=1)
    r, eff = gm.recombination_vec, gm.marker_effects
    base = synthetic_genotypes(1024, M, seed=7).view(np.uint8)
    rng = np.random.default_rng(12)
    pick = rng.integers(0, 1024, n_local, dtype=np.int32)
    eng = HipEngine(r, eff)
    co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    base_p = eng.pack(eng.upload_population(base))
    shard = eng.empty_packed(n_local)
    g = eng.empty((n_local, 1), torch.float32)
    eng.gather_packed(base_p, torch.from_numpy(pick).to(eng.device), n_local, shard, None, g)
    g64 = co.gebv(base, eff, want64=True)[1][pick]
    assert gebv_close(g.cpu().numpy(), g64)
    scores = g[:, 0].contiguous()
    best = eng.topk(scores, k).cpu().numpy()
    assert np.array_equal(best, co.topk(scores.cpu().numpy(), k))  # ties (repeated rows) fall to the lower index
    pool = eng.empty_packed(k)
    eng.gather_packed(shard, torch.from_numpy(best.astype(np.int32)).to(eng.device), k, pool)
    winners = pick[best]  # rows of `base`
    eng.set_cross_window(rank * n_local, n_total)
    key, refkey = co.key_from_seed(11).copy(), co.key_from_seed(11)
    for gen in range(2):
        pairs = rng.integers(0, k, (n_local, 2), dtype=np.int32)
        outp = eng.empty_packed(n_local)
        eng.cross_packed(pool, 1, k, torch.from_numpy(pairs).to(eng.device).reshape(1, -1, 2), n_local, key, 1, outp, None, g)
        keys = co.split(refkey, 2 * n_total + 1)
        refkey = keys[0].copy()
        assert np.array_equal(key, refkey)
        sample = np.unique(np.concatenate(([0, 1, n_local - 1], rng.integers(0, n_local, 45))))
        got = eng.download_population(eng.unpack(outp[torch.from_numpy(sample).to(eng.device)].contiguous())).view(np.uint8)
        ref = np.empty((len(sample), M, 2), np.uint8)
        for j, b in enumerate(sample):
            for p in range(2):
                gk = keys[1 + 2 * (rank * n_local + int(b)) + p]
                mask = co.meiosis_mask(co.split(gk, 2)[0], r)
                ref[j, :, p] = base[winners[pairs[b, p]]][np.arange(M), mask]
        assert np.array_equal(got, ref), f"generation {gen}"
        assert gebv_close(g.cpu().numpy()[sample], co.gebv(ref, eff, want64=True)[1])


@pytest.mark.timeout(900)
def test_c4_whole_population_on_one_device():
    """BASELINE configs[3]'s WHOLE population on one device (288 GB of HBM hold it many times over): 65,536 x 100k as
    bit planes (1.6 GB), its GEBV, the top-6553 of 65,536 with exact ties, the winners' pool, and all 65,536 crosses of
    one split(key, 2 x 65,536 + 1) in ONE call -- the generation the eight shards of the population-sharded flow compute
    between them (tests/test_distributed.py: N ranks == one device), at the largest sizes the BASELINE names.  Two
    generations (the second reads masks made ahead); sampled crosses from both ends, the middle and across the row where
    byte offsets pass 2^32 against the oracle's meiosis for exactly those keys.  The same generation as byte rows is a
    13.1 GB block: its rows on either side of 4 GiB and 8 GiB equal the rows expanded one by one."""
    from breedgym_amd import HipEngine
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    n_total, M, k = 65_536, 100_000, 6553
    gm = synthetic_genetic_map(M, n_chr=21, n_traits=1, seed=1)
    r, eff = gm.recombination_vec, gm.marker_effects
    base = synthetic_genotypes(1024, M, seed=7).view(np.uint8)
    rng = np.random.default_rng(21)
    pick = rng.integers(0, 1024, n_total, dtype=np.int32)
    eng = HipEngine(r, eff)
    co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    base_p = eng.pack(eng.upload_population(base))
    pop = eng.empty_packed(n_total)
    g = eng.empty((n_total, 1), torch.float32)
    eng.gather_packed(base_p, torch.from_numpy(pick).to(eng.device), n_total, pop, None, g)
    g64 = co.gebv(base, eff, want64=True)[1][pick]
    assert gebv_close(g.cpu().numpy(), g64)
    scores = g[:, 0].contiguous()
    best = eng.topk(scores, k).cpu().numpy()
    assert np.array_equal(best, co.topk(scores.cpu().numpy(), k))  # ties (repeated rows) fall to the lower index
    pool = eng.empty_packed(k)
    eng.gather_packed(pop, torch.from_numpy(best.astype(np.int32)).to(eng.device), k, pool)
    winners = pick[best]  # rows of `base`
    del pop
    key, refkey = co.key_from_seed(11).copy(), co.key_from_seed(11)
    row_4g = (1 << 32) // (2 * M)  # the byte-layout row in which offsets pass 4 GiB
    for gen in range(2):
        pairs = rng.integers(0, k, (n_total, 2), dtype=np.int32)
        outp = eng.empty_packed(n_total)
        eng.cross_packed(pool, 1, k, torch.from_numpy(pairs).to(eng.device).reshape(1, -1, 2), n_total, key, 1, outp, None, g)
        keys = co.split(refkey, 2 * n_total + 1)
        refkey = keys[0].copy()
        assert np.array_equal(key, refkey)
        sample = np.unique(np.concatenate(([0, 1, row_4g - 1, row_4g, row_4g + 1, n_total // 2, n_total - 2, n_total - 1],
                                           rng.integers(0, n_total, 24))))
        sample_d = torch.from_numpy(sample).to(eng.device)
        got_rows = eng.unpack(outp[sample_d].contiguous())
        got = eng.download_population(got_rows).view(np.uint8)
        ref = np.empty((len(sample), M, 2), np.uint8)
        for j, b in enumerate(sample):
            for p in range(2):
                gk = keys[1 + 2 * int(b) + p]
                mask = co.meiosis_mask(co.split(gk, 2)[0], r)
                ref[j, :, p] = base[winners[pairs[b, p]]][np.arange(M), mask]
        assert np.array_equal(got, ref), f"generation {gen}"
        assert gebv_close(g.cpu().numpy()[sample], co.gebv(ref, eff, want64=True)[1])
        if gen == 1:
            whole = eng.unpack(outp)  # (65,536, 200,000) bytes: 13.1 GB, offsets to 2^33.6
            assert whole.shape[0] == n_total and whole.numel() > (1 << 33)
            assert torch.equal(whole[sample_d], got_rows)
            gb = eng.empty((n_total, 1), torch.float32)
            eng.gebv(whole, gb)  # the one-pass byte-row GEBV over the same 13.1 GB
            assert gebv_close(gb.cpu().numpy()[sample], co.gebv(ref, eff, want64=True)[1])
            assert torch.equal(eng.pack(whole)[:, : M // 4], outp[:, : M // 4])  # ... and back: the bit planes it came from
            del whole
        del outp


# ---- many long rows: the stream-per-lane mask role + mask_finish_kernel (csrc/mask_stream.h) --------------


@pytest.mark.parametrize("layout", ["original", "partitionable"])
@pytest.mark.parametrize("mode,M", [(0, 16523), (1, 16523), (2, 20000), (0, 50001)])
def test_many_long_rows_stream_mask_role(layout, mode, M, rng):
    """>= 8192 bit-plane rows of > 16384 markers: the launch streams rows in turns, draws the next generation's
    recombination sites one gamete stream per lane (thresholds through the scalar cache) and a second kernel
    turns the site words into masks.  Three generations (the last two read masks made that way), every offspring
    against the oracle; odd M, a half that is no multiple of 64, always-recombining markers, all key modes."""
    from breedgym_amd import HipEngine

    co.set_threefry_layout(layout)
    co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    try:
        E, n, NP = (2, 4100, 37) if M < 30000 else (1, 8200, 29)
        pool = rng.integers(0, 2, (E, NP, M, 2), dtype=np.uint8)
        if mode & 2:
            one = rng.integers(0, NP, (E, n), dtype=np.int32)
            idx = np.stack((one, one), -1)
        else:
            idx = rng.integers(0, NP, (E, n, 2), dtype=np.int32)
        r = (rng.random(M) * 0.004).astype(np.float32)
        r[:: M // 21] = 0.5
        r[rng.random(M) < 0.16] = 0.0
        r[5], r[M // 2 + 3], r[M - 2] = 1.0, 2.0, 1.0
        eff = rng.laplace(0, 6, (M, 1)).astype(np.float32)
        eng = HipEngine(r, eff, threefry_layout=layout)
        packed = eng.pack(eng.upload_population(pool.reshape(E * NP, M, 2)))
        d_idx = torch.from_numpy(idx).to(eng.device)
        key, ref_key = co.key_from_seed(31).copy(), co.key_from_seed(31)
        for gen in range(4):
            eng.set_mask_prefetch(gen != 3)  # the last generation makes its masks in stream order (mask_kernel<4>)
            outp = eng.empty_packed(E * n)
            ob = eng.empty_population(E * n) if gen != 1 else None
            g = eng.empty((E, n, 1), torch.float32)
            eng.cross_packed(packed, E, NP, d_idx, n, key, mode, outp, ob, g)
            ref, ref_key = co.cross(pool, idx, r, ref_key, mode)
            got = eng.download_population(eng.unpack(outp)).view(np.uint8).reshape(ref.shape)
            assert np.array_equal(got, ref), f"gen {gen}: {np.argwhere(got != ref)[:4]}"
            if ob is not None:
                assert np.array_equal(eng.download_population(ob).view(np.uint8).reshape(ref.shape), ref)
            assert np.array_equal(key, ref_key)
            assert gebv_close(g.cpu().numpy(), co.gebv(ref, eff, want64=True)[1])
        eng.sync()
    finally:
        co.set_threefry_layout("original")


@pytest.mark.parametrize("M", [16385, 17000, 17920])
def test_dma_stream_short_last_chunks_soak(M, rng):
    """Rows of exactly five 4096-marker chunks whose last chunk fills at most three of its eight slices
    (16384 < M <= 17920): the shape in which the LDS-DMA stream's counted `vmcnt` wait used to lean on timing (a
    store with no active lane is never issued, hence never counted; csrc/packed_dma.h now sends those lanes to a
    spare line).  8200 rows, 50 cross launches down the key chain (observation written on even launches, not on odd
    ones) and 50 founder gathers, every launch compared on the device with the oracle's rows / the gathered rows."""
    from breedgym_amd import HipEngine

    co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    E, n, NP = 2, 4100, 23
    pool = rng.integers(0, 2, (E, NP, M, 2), dtype=np.uint8)
    r = (rng.random(M) * 0.004).astype(np.float32)
    r[:: M // 21] = 0.5
    eff = rng.laplace(0, 6, (M, 1)).astype(np.float32)
    eng = HipEngine(r, eff)

    def packed_rows(g):  # (rows, M, 2) bytes -> the library's bit-plane rows (include/breedhip.h), on the device
        bits = np.packbits(g.reshape(g.shape[0], 2 * M), axis=1, bitorder="little")
        out = np.zeros((g.shape[0], eng.packed_pitch), np.uint8)
        out[:, : bits.shape[1]] = bits
        return torch.from_numpy(out).to(eng.device)

    packed = eng.pack(eng.upload_population(pool.reshape(E * NP, M, 2)))
    assert torch.equal(packed, packed_rows(pool.reshape(E * NP, M, 2)))
    key, ref_key = co.key_from_seed(77).copy(), co.key_from_seed(77)
    for it in range(50):
        idx = rng.integers(0, NP, (E, n, 2), dtype=np.int32)
        outp = eng.empty_packed(E * n)
        ob = eng.empty_population(E * n) if it % 2 == 0 else None
        eng.cross_packed(packed, E, NP, torch.from_numpy(idx).to(eng.device), n, key, 0, outp, ob, None)
        ref, ref_key = co.cross(pool, idx, r, ref_key, 0)
        assert torch.equal(outp, packed_rows(ref.reshape(E * n, M, 2))), f"cross launch {it}"
        if ob is not None:
            assert torch.equal(ob, eng.unpack(outp)), f"observation of cross launch {it}"
        assert np.array_equal(key, ref_key)
    src = packed_rows(rng.integers(0, 2, (300, M, 2), dtype=np.uint8))
    for it in range(50):
        gi = torch.from_numpy(rng.integers(0, 300, 8200).astype(np.int32)).to(eng.device)
        outp = eng.empty_packed(8200)
        ob = eng.empty_population(8200) if it % 2 == 0 else None
        eng.gather_packed(src, gi, 8200, outp, ob)
        assert torch.equal(outp, src[gi.long()]), f"gather launch {it}"
        if ob is not None:
            assert torch.equal(ob, eng.unpack(outp)), f"observation of gather launch {it}"
    eng.sync()


def test_c1_single_env_at_shape():
    """BASELINE configs[0] at its real shape on the device: single-env `BreedGym`, 370 individuals x 10k markers,
    16 random crosses per generation, 10 generations, the reward (mean GEBV, breedgym.py:151-155) at the horizon --
    the loop of the reference's tests/test_env.py:103-119 -- every population byte for byte and every GEBV against
    the chain founders -> co.cross -> co.cross ...; then a second episode after reset()."""
    from breedgym_amd import BreedGym
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    N, M, C, G = 370, 10_000, 16, 10
    gm = synthetic_genetic_map(M, n_chr=10, seed=1)
    geno = synthetic_genotypes(N, M, seed=0)
    env = BreedGym(initial_population=geno, genetic_map=gm, reward_shaping=False, num_generations=G)
    assert type(env.simulator.engine).__name__ == "HipEngine"
    arng = np.random.default_rng(2)
    key = None
    for episode, seed in enumerate((7, None)):
        pop, info = env.reset(seed=seed) if seed is not None else env.reset()
        assert pop.shape == (N, M, 2) and np.array_equal(u8(pop), geno.view(np.uint8))
        if seed is not None:
            key = co.key_from_seed(seed)  # reset(seed) re-seeds the simulator (breedgym.py:117); reset() keeps the chain
        ref = geno.view(np.uint8)[None]
        for g in range(G):
            action = arng.integers(0, ref.shape[1], (C, 2))
            pop, rew, ter, tru, info = env.step(action)
            ref, key = co.cross(ref, action[None].astype(np.int32), gm.recombination_vec, key, 0)
            assert pop.shape == (C, M, 2) and np.array_equal(u8(pop), ref[0]), (episode, g)
            g64 = co.gebv(ref[0], gm.marker_effects, want64=True)[1]
            assert gebv_close(info["GEBV"].to_numpy(), g64)
            assert np.array_equal(env.simulator.random_key, key)
            assert not ter and tru == (g == G - 1)
            if g < G - 1:
                assert rew == 0
            else:
                assert abs(rew - g64.mean()) <= 1e-5 * max(1.0, abs(g64.mean()))


# ---- rows beyond one workgroup's LDS: site words through memory (csrc/mask_stream.h, mask_sites_kernel) ----------


@pytest.mark.parametrize("layout,M,C,mode", [("original", 1_406_757, 300, 1), ("partitionable", 1_406_757, 64, 0),
                                              ("original", 262_145, 500, 1), ("original", 300_001, 130, 3)])
def test_rows_beyond_the_lds_mask_limit(layout, M, C, mode, rng):
    """The reference's own cross timing shape (scripts/time_cross.py:11-14: 300 progenies x 1,406,757 markers) and
    rows just past 2^18 markers: every offspring byte of two calls against the oracle, bit planes and byte layout,
    GEBV within 1e-5 -- the prefix-XOR of a row runs over site words in memory with the crossover parity carried
    from tile to tile."""
    from breedgym_amd import HipEngine, _lib

    assert _lib.lib().bh_max_markers() >= 1 << 21
    co.set_threefry_layout(layout)
    co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    try:
        NP = 24
        pool = rng.integers(0, 2, (1, NP, M, 2), dtype=np.uint8)
        r = (rng.random(M) * 0.0004).astype(np.float32)
        r[:: M // 21] = 0.5
        r[rng.random(M) < 0.16] = 0.0
        r[7], r[M // 2 + 1], r[M - 1] = 1.0, 1.5, 1.0
        eff = rng.laplace(0, 6, (M, 1)).astype(np.float32)
        eng = HipEngine(r, eff, threefry_layout=layout)
        if mode & 2:
            one = rng.integers(0, NP, (1, C), dtype=np.int32)
            idx = np.stack((one, one), -1)
        else:
            idx = rng.integers(0, NP, (1, C, 2), dtype=np.int32)
        d_idx = torch.from_numpy(idx).to(eng.device)
        rows = eng.upload_population(pool[0])
        packed = eng.pack(rows)
        key_p, key_b, ref_key = co.key_from_seed(5).copy(), co.key_from_seed(5).copy(), co.key_from_seed(5)
        for call in range(2):
            ref, ref_key = co.cross(pool, idx, r, ref_key, mode)
            g64 = co.gebv(ref, eff, want64=True)[1]
            outp, ob, g = eng.empty_packed(C), eng.empty_population(C), eng.empty((1, C, 1), torch.float32)
            eng.cross_packed(packed, 1, NP, d_idx, C, key_p, mode, outp, ob, g)
            assert np.array_equal(eng.download_population(ob).view(np.uint8).reshape(ref.shape), ref), f"packed, call {call}"
            assert torch.equal(eng.unpack(outp), ob)
            assert np.array_equal(key_p, ref_key) and gebv_close(g.cpu().numpy(), g64)
            if call == 0:  # the byte-layout state takes the same masks through its own row kernel
                out, g2 = eng.empty_population(C), eng.empty((1, C, 1), torch.float32)
                eng.cross(rows, 1, NP, d_idx, C, key_b, mode, out, g2)
                assert torch.equal(out, ob) and np.array_equal(key_b, ref_key) and gebv_close(g2.cpu().numpy(), g64)
        eng.sync()
    finally:
        co.set_threefry_layout("original")


@pytest.mark.parametrize("rows,M,T", [(340, 100_003, 8), (4100, 8_200, 1), (129, 262_144 + 8, 10), (2048, 16_400, 17),
                                        (5000, 7_001, 3), (32, 1_048_583, 2)])
def test_byte_row_gebv_in_one_pass(rows, M, T, rng):
    """bh_gebv on large byte-layout populations (csrc/gebv_bytes.h: bit planes built in LDS, digit GEMMs on the matrix
    cores): ragged row counts (tiles and row groups that reach past the last row), marker counts that are no multiple
    of 8 / 512 (a last vector with pad bytes, a last chunk with missing vectors), one and two column tiles, more than
    one pack of traits; effects with large and tiny magnitudes side by side."""
    from breedgym_amd import HipEngine

    assert rows * M >= 1 << 25
    co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    eff = (rng.laplace(0, 6, (M, T)) * 10.0 ** rng.integers(-3, 3, (1, T))).astype(np.float32)
    eng = HipEngine(np.full(M, 0.01, np.float32), eff)
    geno = (rng.random((rows, M, 2)) < 0.3).astype(np.uint8)
    geno[0], geno[-1] = 1, 0  # the largest and the smallest sums
    got = eng.gebv(eng.upload_population(geno)).cpu().numpy()
    assert gebv_close(got, co.gebv(geno, eff, want64=True)[1])


@pytest.mark.parametrize("T", [1, 6, 10, 13])
def test_gebv_passes_leave_their_accumulator_clean(T, rng):
    """The matrix-core GEBV passes add 64-bit sums into one accumulator per context and count arrivals per row group
    (csrc/gebv_bits.h: gebv_tile_out, gebv_group_done); the last workgroup of a row group leaves both at zero.  Calls
    of different sizes, through both passes (bit planes, byte rows), growing and shrinking, one after the other on the
    same context: every one equals the oracle.  T = 6: trait 5 lies across the two column tiles; T = 13: two packs."""
    from breedgym_amd import HipEngine

    M = 20_011
    co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    eff = (rng.laplace(0, 6, (M, T)) * 10.0 ** rng.integers(-2, 2, (1, T))).astype(np.float32)
    eng = HipEngine(np.full(M, 0.01, np.float32), eff)
    geno = (rng.random((4000, M, 2)) < 0.35).astype(np.uint8)
    g64 = co.gebv(geno, eff, want64=True)[1]
    rows = eng.upload_population(geno)
    packed = eng.pack(rows)
    for n, route in ((100, "bits"), (1000, "bits"), (37, "bits"), (4000, "bytes"), (129, "bits"), (4000, "bits"), (2000, "bytes"),
                     (1, "bits")):
        if route == "bits":
            g = eng.empty((n, T), torch.float32)
            eng.gather_packed(packed[:n].contiguous(), None, n, None, None, g)
        else:
            assert n * M >= 1 << 25
            g = eng.gebv(rows[:n].contiguous())
        assert gebv_close(g.cpu().numpy(), g64[:n]), (n, route)


def test_gebv_passes_on_random_shapes():
    """A seeded sweep of shapes through the matrix-core GEBV passes -- row counts that are no multiple of a tile or a
    row group, marker counts that end a row inside a macro-chunk, any number of traits up to two packs: the K slicing
    (even slices, whole eights per XCD round), the padded rows and the unused trait slots of the accumulator must never
    show.  Bit planes always; byte rows where the population is large enough for the one-pass kernel."""
    from breedgym_amd import HipEngine

    co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    rng = np.random.default_rng(20240607)
    for case in range(24):
        T = int(rng.integers(1, 13))
        rows = int(rng.integers(1, 900)) if case % 3 else int(rng.integers(900, 2600))
        M = int(rng.integers(16_385, 60_000))
        eff = (rng.laplace(0, 4, (M, T)) * 10.0 ** rng.integers(-2, 3, (1, T))).astype(np.float32)
        eng = HipEngine(np.full(M, 0.01, np.float32), eff)
        geno = (rng.random((rows, M, 2)) < rng.uniform(0.05, 0.6)).astype(np.uint8)
        g64 = co.gebv(geno, eff, want64=True)[1]
        d = eng.upload_population(geno)
        g = eng.empty((rows, T), torch.float32)
        eng.gather_packed(eng.pack(d), None, rows, None, None, g)
        assert gebv_close(g.cpu().numpy(), g64), ("bit planes", rows, M, T)
        assert gebv_close(eng.gebv(d).cpu().numpy(), g64), ("byte rows", rows, M, T, rows * M >= 1 << 25)
        del eng


@pytest.mark.parametrize("E,n", [(1, 1), (3, 7), (4, 200), (2, 1000), (1, 1500)])
def test_pair_actions_kernel(E, n, rng):
    """bh_pair_actions (PairScores._convert_actions in one kernel behind the top-k) against the NumPy restatement of
    vec_wrappers.py:100-112, incl. rows longer than one workgroup sorts (n * n > 16384) and exact score ties"""
    from breedgym_amd import HipEngine

    eng = HipEngine(np.full(8, 0.1, np.float32), np.ones((8, 1), np.float32))
    scores = rng.standard_normal((E, n, n)).astype(np.float32)
    if n > 2:
        scores[:, 1, 2] = scores[:, 0, 1] = scores.max() + 1.0  # two pairs tie for the best score
    got = eng.pair_actions(torch.from_numpy(scores), n).cpu().numpy()
    assert np.array_equal(got, _pair_actions_restated(scores, n))


def test_pair_actions_of_a_multi_trait_outer_sum(rng):
    """What the reference's own PairScores test feeds with a 7-trait map (tests/test_vec.py:196-200:
    np.add.outer of (E, n, T) GEBVs -> (E, n, T, E, n, T)): `_convert_actions` flattens an env's block whatever its
    shape (vec_wrappers.py:102) and the pairs are (flat // n, flat % n), beyond n - 1 and all.  bh_pair_actions takes
    the row length; the env then clamps the indices like JAX's gather."""
    from breedgym_amd import HipEngine

    E, n, T = 2, 24, 3
    eng = HipEngine(np.full(8, 0.1, np.float32), np.ones((8, 1), np.float32))
    g = rng.standard_normal((E, n, T)).astype(np.float32)
    scores = np.add.outer(g, g)  # (E, n, T, E, n, T)
    got = eng.pair_actions(torch.from_numpy(scores), n).cpu().numpy()
    ref = _pair_actions_restated(scores.reshape(E, -1), n)
    assert np.array_equal(got, ref) and ref.max() >= n  # indices beyond the generation do occur


# ---- the reset gathered by the horizon step's launch (bh_gather_packed_deferred) -----------------------


@pytest.mark.parametrize("E,n,M,T,G,observations", [
    (3, 24, 2000, 1, 3, "eager"),      # few rows: one row per wave
    (16, 256, 1000, 2, 2, "eager"),    # 4096 rows: two rows per wave, two traits, the draw rides on step 1 of 2
    (5, 40, 1237, 1, 4, "lazy"),       # no byte observation, ragged row length
    (4, 33, 520, 2, 1, "eager"),       # every step is a horizon: nothing can be made a step ahead
    (2, 16, 300, 3, 3, "eager"),       # three traits: the GEBV is a pass of its own, the gather gets a launch of its own
])
def test_reset_made_by_the_horizon_launch_equals_oracle_chain(E, n, M, T, G, observations):
    from breedgym_amd import VecBreedGym
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    N = max(50, n + 7)
    gm = synthetic_genetic_map(M, n_chr=3, n_traits=T, seed=4)
    geno = synthetic_genotypes(N, M, seed=6)
    env = VecBreedGym(num_envs=E, initial_population=geno, individual_per_gen=n, num_generations=G,
                      reward_shaping=True, genetic_map=gm, observations=observations)
    pop, info = env.reset(seed=9)
    f, env_key = co.founders(co.key_from_seed(9), E, N, n)
    ref = geno.view(np.uint8)[f]
    sim_key = co.key_from_seed(9)
    arng = np.random.default_rng(8)
    kept = []  # what a caller keeps must stay what it was (the ring hands the horizon's blocks out of three slots)
    for step in range(3 * G + 1):  # three horizons
        a = arng.integers(0, n, (E, n, 2), dtype=np.int32)
        pop, rews, ter, tru, info = env.step(a)
        ref, sim_key = co.cross(ref, a, gm.recombination_vec, sim_key, 0)
        g64 = co.gebv(ref, gm.marker_effects, want64=True)[1]
        assert gebv_close(np.asarray(info["GEBV"]), g64), f"step {step}"
        assert np.allclose(rews, g64.max(axis=(1, 2)), rtol=GEBV_RTOL)
        assert np.array_equal(env.simulator.random_key, sim_key)
        if (step + 1) % G == 0:
            assert tru.all() and env.step_idx == 0
            f, env_key = co.founders(env_key, E, N, n)
            ref = geno.view(np.uint8)[f]
            assert gebv_close(np.asarray(env.reset_infos["GEBV"]), co.gebv(ref, gm.marker_effects, want64=True)[1])
        assert np.array_equal(u8(pop), ref), f"step {step}"
        assert np.array_equal(env.random_key, env_key)
        assert np.array_equal(u8(env.simulator.engine.as_genotypes(env.simulator.engine.unpack(env._prows), E, n)), ref)
        kept.append((pop, ref.copy(), info["GEBV"], g64))
    for pop, r, g, g64 in kept[-4:]:
        assert np.array_equal(u8(pop), r)
        assert gebv_close(np.asarray(g), g64)


def test_reset_made_ahead_is_dropped_when_the_key_or_the_seed_changes():
    """Founders drawn a step ahead belong to the env key they were drawn from: a wrapper that advances the key per step
    (SelectionScores), a reset(seed=...) by hand and set_state all void them; results stay the oracle's."""
    from breedgym_amd import VecBreedGym, rng
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    E, n, M, N, G = 4, 32, 800, 40, 3
    gm = synthetic_genetic_map(M, n_chr=2, n_traits=1, seed=4)
    geno = synthetic_genotypes(N, M, seed=6)
    env = VecBreedGym(num_envs=E, initial_population=geno, individual_per_gen=n, num_generations=G,
                      reward_shaping=True, genetic_map=gm)
    env.reset(seed=1)
    f, env_key = co.founders(co.key_from_seed(1), E, N, n)
    ref = geno.view(np.uint8)[f]
    sim_key = co.key_from_seed(1)
    arng = np.random.default_rng(3)
    for step in range(2 * G):
        a = arng.integers(0, n, (E, n, 2), dtype=np.int32)
        # somebody advances the env key before every step, as SelectionScores' shuffle does
        env.random_key = rng.split(env.random_key, 2, env.simulator.threefry_layout)[0]
        env_key = co.split(env_key, 2)[0]
        pop, rews, ter, tru, info = env.step(a)
        ref, sim_key = co.cross(ref, a, gm.recombination_vec, sim_key, 0)
        if (step + 1) % G == 0:
            f, env_key = co.founders(env_key, E, N, n)
            ref = geno.view(np.uint8)[f]
        assert np.array_equal(u8(pop), ref), f"step {step}"
        assert np.array_equal(env.random_key, env_key)
    # a reset by hand with a seed, one step before a horizon (founders are drawn ahead by then)
    for step in range(G - 1):
        env.step(arng.integers(0, n, (E, n, 2), dtype=np.int32))
    assert env._founders_ahead is not None
    pop, _ = env.reset(seed=77)
    f, env_key = co.founders(co.key_from_seed(77), E, N, n)
    assert np.array_equal(u8(pop), geno.view(np.uint8)[f])
    assert np.array_equal(env.random_key, env_key)


def test_reset_with_a_new_population_size_crosses_everything_made_ahead():
    """The reference's ``reset(options={"individual_per_gen": n})`` (vec_env.py:118-119) in the middle of an episode, one
    step before a horizon -- when the founders of the next reset are drawn ahead for the OLD size, and again right
    behind a horizon step whose launch gathered the next episode's founders ahead: both must be dropped (sizes differ),
    and everything after follows the oracle's chain at the new size, through its own horizons."""
    from breedgym_amd import VecBreedGym
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    E, n0, M, N, G = 4, 32, 800, 40, 3
    gm = synthetic_genetic_map(M, n_chr=2, n_traits=1, seed=4)
    geno = synthetic_genotypes(N, M, seed=6)
    env = VecBreedGym(num_envs=E, initial_population=geno, individual_per_gen=n0, num_generations=G, reward_shaping=True,
                      genetic_map=gm)
    env.reset(seed=5)
    f, env_key = co.founders(co.key_from_seed(5), E, N, n0)
    ref, sim_key = geno.view(np.uint8)[f], co.key_from_seed(5)
    arng = np.random.default_rng(8)

    def steps(k, n):
        nonlocal ref, sim_key, env_key
        for _ in range(k):
            a = arng.integers(0, n, (E, n, 2), dtype=np.int32)
            pop, rews, ter, tru, info = env.step(a)
            ref, sim_key = co.cross(ref, a, gm.recombination_vec, sim_key, 0)
            g = co.gebv(ref, gm.marker_effects)
            assert np.allclose(np.asarray(info["GEBV"]), g, rtol=1e-5, atol=1e-5)
            if env.step_idx == 0:  # the horizon: the observation is the next episode's founders
                f, env_key = co.founders(env_key, E, N, n)
                ref = geno.view(np.uint8)[f]
            assert np.array_equal(u8(pop), ref) and np.array_equal(env.random_key, env_key)

    steps(G - 1, n0)  # one step before the horizon: the founders of the next reset are drawn (for 32 per env)
    assert env._founders_ahead is not None and env._founders_ahead[0].numel() == E * n0
    for n in (24, 40):  # smaller, then larger than what was made ahead
        pop, info = env.reset(options={"individual_per_gen": n})
        f, env_key = co.founders(env_key, E, N, n)
        ref = geno.view(np.uint8)[f]
        assert pop.shape == (E, n, M, 2) and np.array_equal(u8(pop), ref) and np.array_equal(env.random_key, env_key)
        assert np.allclose(np.asarray(info["GEBV"]), co.gebv(ref, gm.marker_effects), rtol=1e-5, atol=1e-5)
        steps(2 * G + 1, n)  # two horizons at the new size (their founders made ahead again) and one step more
        assert env.step_idx == 1
        steps(G - 1, n)      # ... to right behind the next horizon: its launch gathered founders ahead, now for `n`
        assert env.step_idx == 0
