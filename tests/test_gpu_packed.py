"""GPU: bit-plane (1 bit per allele) rows through the C ABI -- pack / unpack, the packed cross and
gather against the CPU oracle and against the byte-layout kernels: bit-exact genotypes, GEBV within
1e-5, identical key chains."""
import numpy as np
import pytest
import torch
from golden_util import CASES, load_case
from oracle_engine import packed_pitch_of

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu

GEBV_RTOL = 1e-5


def make_engine(r, eff):
    from breedgym_amd import HipEngine

    return HipEngine(r, eff)


def gebv_close(got, ref64):
    scale = np.maximum(np.abs(ref64), 1.0)
    return np.all(np.abs(got.astype(np.float64) - ref64) <= GEBV_RTOL * scale)


def pack_numpy(g):
    """(rows, M, 2) 0/1 -> (rows, packed pitch / 2) uint16 in the documented layout (include/breedhip.h): bit i of
    word v = byte i of the 16-byte vector v of the byte-layout row."""
    rows, M, _ = g.shape
    by = np.zeros((rows, packed_pitch_of(M) * 8), np.uint8)
    by[:, : 2 * M] = g.reshape(rows, 2 * M)
    return np.packbits(by, axis=1, bitorder="little").view(np.uint16)


def packed_host(t):
    return t.cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("M", [1, 5, 8, 9, 63, 64, 65, 100, 1000, 1001, 10000, 16384, 16385, 16896, 20000, 32768 + 64])
def test_pack_unpack_roundtrip(M, rng):
    N = 7
    g = rng.integers(0, 2, (N, M, 2), dtype=np.uint8)
    eng = make_engine(np.full(M, 0.01, np.float32), np.ones((M, 1), np.float32))
    rows = eng.upload_population(g)
    pk = eng.pack(rows)
    assert pk.shape == (N, eng.packed_pitch) and eng.packed_pitch == packed_pitch_of(M)
    assert np.array_equal(packed_host(pk), pack_numpy(g))
    back = eng.unpack(pk)
    assert np.array_equal(eng.download_population(back).view(np.uint8), g)
    # the padding of the byte rows is zero too (they are read 16 bytes at a time)
    assert np.array_equal(back.cpu().numpy()[:, 2 * M:], np.zeros((N, eng.pitch - 2 * M), np.uint8))


def run_packed_cross(eng, pool, idx, seed, mode, want_bytes=True, want_gebv=True, reward=False):
    E, n_pool, M, _ = pool.shape
    n_out = idx.shape[1]
    ppool = eng.pack(eng.upload_population(pool.reshape(E * n_pool, M, 2)))
    outp = eng.empty_packed(E * n_out)
    outb = eng.empty_population(E * n_out) if want_bytes else None
    gebv = eng.empty((E, n_out, eng.n_traits), torch.float32) if want_gebv else None
    rew = eng.empty((E,), torch.float32) if reward else None
    key = co.key_from_seed(seed).copy()
    eng.cross_packed(ppool, E, n_pool, torch.from_numpy(idx).to(eng.device), n_out, key, mode, outp, outb, gebv, rew)
    got_b = eng.download_population(outb).view(np.uint8).reshape(E, n_out, M, 2) if want_bytes else None
    return packed_host(outp), got_b, key, (gebv.cpu().numpy() if want_gebv else None), (
        rew.cpu().numpy() if reward else None)


@pytest.mark.parametrize("name", CASES)
def test_golden_fixtures_packed(name):
    c = load_case(name)
    eng = make_engine(c["r"], c["effects"])
    pk, got, key, g, _ = run_packed_cross(eng, c["pool"], c["idx"], c["seed"], c["rng_mode"])
    assert np.array_equal(got, c["out"]), "byte-layout offspring differ from the golden fixture"
    E, n_out, M, _ = c["out"].shape
    assert np.array_equal(pk, pack_numpy(c["out"].reshape(E * n_out, M, 2)))
    assert np.array_equal(key, c["new_key"])
    assert gebv_close(g, c["gebv64"])


@pytest.mark.parametrize("M", [1, 2, 7, 8, 9, 63, 64, 65, 127, 128, 129, 1000, 1001, 4999, 10000])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_packed_cross_vs_oracle(M, mode, rng):
    E, NP, NO, T = 3, 11, 13, 2
    pool = rng.integers(0, 2, (E, NP, M, 2), dtype=np.uint8)
    idx = rng.integers(-NP, NP + 2, (E, NO, 2), dtype=np.int32)
    r = (rng.random(M) * 0.05).astype(np.float32)
    r[:: max(1, M // 5)] = 0.5
    r[rng.random(M) < 0.16] = 0.0
    eff = rng.laplace(0, 6, (M, T)).astype(np.float32)
    seed = int(rng.integers(0, 2**32))
    eng = make_engine(r, eff)
    want_bytes = (M + mode) % 2 == 0  # with and without the byte-layout observation
    pk, got, key, g, rew = run_packed_cross(eng, pool, idx, seed, mode, want_bytes=want_bytes, reward=True)
    ref, refkey = co.cross(pool, idx, r, co.key_from_seed(seed), mode)
    if want_bytes:
        assert np.array_equal(got, ref), f"M={M} mode={mode}: {np.argwhere(got != ref)[:5]}"
    assert np.array_equal(pk, pack_numpy(ref.reshape(E * NO, M, 2)))
    assert np.array_equal(key, refkey)
    _, g64 = co.gebv(ref, eff, want64=True)
    assert gebv_close(g, g64)
    assert np.array_equal(rew, g.reshape(E, -1).max(axis=1))


@pytest.mark.parametrize("M", [16392, 50001])
def test_packed_cross_long_rows(M, rng):
    E, NP, NO = 2, 5, 6
    pool = rng.integers(0, 2, (E, NP, M, 2), dtype=np.uint8)
    idx = rng.integers(0, NP, (E, NO, 2), dtype=np.int32)
    r = (rng.random(M) * 0.002).astype(np.float32)
    r[:: M // 7] = 0.5
    eff = rng.laplace(0, 6, (M, 3)).astype(np.float32)
    eng = make_engine(r, eff)
    for mode in (0, 1):
        pk, got, key, g, _ = run_packed_cross(eng, pool, idx, 42, mode)
        ref, refkey = co.cross(pool, idx, r, co.key_from_seed(42), mode)
        assert np.array_equal(got, ref) and np.array_equal(key, refkey)
        assert np.array_equal(pk, pack_numpy(ref.reshape(E * NO, M, 2)))
        assert gebv_close(g, co.gebv(ref, eff, want64=True)[1])


@pytest.mark.parametrize("T", [1, 2, 3, 4, 8, 9, 17])
@pytest.mark.parametrize("want_bytes", [False, True])
def test_packed_traits(T, want_bytes, rng):
    """fused f64 GEBV (T <= 8), matrix-core pass over the byte rows (T >= 3 with observations),
    chained f64 passes over the packed rows (T > 8 without)."""
    M, N, NO = 1000, 40, 37
    pop = rng.integers(0, 2, (1, N, M, 2), dtype=np.uint8)
    eff = rng.laplace(0, 6, (M, T)).astype(np.float32)
    eff[3, 0] = 475.0
    eng = make_engine(np.full(M, 0.01, np.float32), eff)
    idx = rng.integers(0, N, (1, NO, 2), dtype=np.int32)
    pk, got, _, g, _ = run_packed_cross(eng, pop, idx, 3, 0, want_bytes=want_bytes)
    ref, _ = co.cross(pop, idx, np.full(M, 0.01, np.float32), co.key_from_seed(3), 0)
    assert np.array_equal(pk, pack_numpy(ref.reshape(NO, M, 2)))
    assert gebv_close(g[0], co.gebv(ref[0], eff, want64=True)[1])
    # gather + GEBV through an index, and GEBV of packed rows in place
    ppool = eng.pack(eng.upload_population(pop[0]))
    gi = torch.from_numpy(rng.integers(-N, N + 2, 33, dtype=np.int32)).to(eng.device)
    outp = eng.empty_packed(33)
    outb = eng.empty_population(33) if want_bytes else None
    gg = eng.empty((33, T), torch.float32)
    eng.gather_packed(ppool, gi, 33, outp, outb, gg)
    ii = gi.cpu().numpy().astype(np.int64)
    ii = np.clip(np.where(ii < 0, ii + N, ii), 0, N - 1)
    sel = pop[0][ii]
    assert np.array_equal(packed_host(outp), pack_numpy(sel))
    if want_bytes:
        assert np.array_equal(eng.download_population(outb).view(np.uint8), sel)
    assert gebv_close(gg.cpu().numpy(), co.gebv(sel, eff, want64=True)[1])
    g2 = eng.empty((N, T), torch.float32)
    eng.gather_packed(ppool, None, N, None, None, g2)
    assert gebv_close(g2.cpu().numpy(), co.gebv(pop[0], eff, want64=True)[1])


@pytest.mark.parametrize("M,T,want_rows", [(20000, 1, False), (20000, 1, True), (1000, 3, False), (40000, 8, True)])
def test_gebv_of_many_rows_out_of_few(M, T, want_rows, rng):
    """A gather of many rows out of a small pool with the GEBV on the matrix cores (a reset's founders at configs[2]:
    4 x 2048 out of 2048): the pass runs over the POOL and the values go through the index (gebv.hip, gebv_bits) --
    indices that wrap and clamp included; with and without the rows themselves."""
    N, R = 40, 173
    pop = rng.integers(0, 2, (N, M, 2), dtype=np.uint8)
    eff = rng.laplace(0, 6, (M, T)).astype(np.float32)
    eng = make_engine(np.full(M, 0.01, np.float32), eff)
    ppool = eng.pack(eng.upload_population(pop))
    gi = torch.from_numpy(rng.integers(-N - 1, N + 2, R, dtype=np.int32)).to(eng.device)
    ii = gi.cpu().numpy().astype(np.int64)
    ii = np.clip(np.where(ii < 0, ii + N, ii), 0, N - 1)
    outp = eng.empty_packed(R) if want_rows else None
    gg = eng.empty((R, T), torch.float32)
    eng.gather_packed(ppool, gi, R, outp, None, gg)
    assert gebv_close(gg.cpu().numpy(), co.gebv(pop[ii], eff, want64=True)[1])
    if want_rows:
        assert np.array_equal(packed_host(outp), pack_numpy(pop[ii]))
    # ... and the same values as the pass over the gathered rows themselves gives, to the bit (three traits and more:
    # both are the matrix-core pass)
    if want_rows and T >= 3:
        g2 = eng.empty((R, T), torch.float32)
        eng.gather_packed(outp, None, R, None, None, g2)
        assert torch.equal(g2, gg)


def test_packed_equals_byte_path_over_generations(rng):
    """Ten generations, both layouts side by side from the same seed: same offspring, same keys, with
    the run-ahead masks of one layout's launches feeding the next call of the same layout."""
    E, n, M, T = 4, 64, 1500, 1
    r = (rng.random(M) * 0.01).astype(np.float32)
    r[::300] = 0.5
    eff = rng.laplace(0, 6, (M, T)).astype(np.float32)
    eng_b, eng_p = make_engine(r, eff), make_engine(r, eff)
    pop = rng.integers(0, 2, (E * n, M, 2), dtype=np.uint8)
    rows_b = eng_b.upload_population(pop)
    rows_p = eng_p.pack(eng_p.upload_population(pop))
    kb, kp = co.key_from_seed(9).copy(), co.key_from_seed(9).copy()
    for gen in range(10):
        idx = torch.from_numpy(rng.integers(0, n, (E, n, 2), dtype=np.int32)).cuda()
        out_b = eng_b.empty_population(E * n)
        gb = eng_b.empty((E, n, T), torch.float32)
        eng_b.cross(rows_b, E, n, idx, n, kb, gen % 2, out_b, gb)
        out_p, obs = eng_p.empty_packed(E * n), eng_p.empty_population(E * n)
        gp = eng_p.empty((E, n, T), torch.float32)
        eng_p.cross_packed(rows_p, E, n, idx, n, kp, gen % 2, out_p, obs, gp)
        assert torch.equal(obs, out_b) and np.array_equal(kb, kp)
        assert torch.equal(eng_p.unpack(out_p), out_b)
        assert torch.allclose(gb, gp, rtol=1e-6, atol=1e-4)
        rows_b, rows_p = out_b, out_p


def test_packed_errors_and_empty(rng):
    M = 100
    eng = make_engine(np.full(M, 0.01, np.float32), np.ones((M, 1), np.float32))
    pk = eng.pack(eng.upload_population(rng.integers(0, 2, (4, M, 2), dtype=np.uint8)))
    key = co.key_from_seed(0).copy()
    idx = torch.zeros((1, 2, 2), dtype=torch.int32, device=eng.device)
    with pytest.raises(ValueError):  # output aliases the pool
        eng.cross_packed(pk, 1, 4, idx, 2, key, 0, pk)
    with pytest.raises(ValueError):
        eng.cross_packed(pk, 1, 4, idx, 2, key, 7, eng.empty_packed(2))
    # zero crosses: the key still advances like split(key, 1)[0]
    k0 = key.copy()
    eng.cross_packed(pk, 1, 4, torch.zeros((1, 0, 2), dtype=torch.int32, device=eng.device), 0, key, 0,
                     eng.empty_packed(1))
    assert np.array_equal(key, co.split(k0, 1)[0])
    with pytest.raises(ValueError):  # identity gather of more rows than there are
        eng.gather_packed(pk, None, 5, eng.empty_packed(5))


def test_packed_full_size_properties():
    """BASELINE configs[1] size: closure and agreement with the byte-layout kernel on device."""
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    gm = synthetic_genetic_map(10_000)
    E, n, M = 16, 256, 10_000
    g = synthetic_genotypes(370, M)
    rng = np.random.default_rng(5)
    from breedgym_amd import Simulator

    sim = Simulator(genetic_map=gm, seed=0)
    eng = sim.engine
    founders = torch.from_numpy(rng.integers(0, 370, E * n, dtype=np.int32)).cuda()
    germ = eng.upload_population(g)
    rows = eng.empty_population(E * n)
    eng.gather(germ, founders, rows)
    prow = eng.empty_packed(E * n)
    obs0 = eng.empty_population(E * n)
    eng.gather_packed(eng.pack(germ), founders, E * n, prow, obs0)
    assert torch.equal(obs0, rows)
    idx = torch.from_numpy(rng.integers(0, n, (E, n, 2), dtype=np.int32)).cuda()
    for mode in (0, 1):
        kb, kp = co.key_from_seed(1).copy(), co.key_from_seed(1).copy()
        out_b, gb = eng.empty_population(E * n), eng.empty((E, n, 1), torch.float32)
        eng.cross(rows, E, n, idx, n, kb, mode, out_b, gb)
        out_p, obs, gp = eng.empty_packed(E * n), eng.empty_population(E * n), eng.empty((E, n, 1), torch.float32)
        rw = eng.empty((E,), torch.float32)
        eng.cross_packed(prow, E, n, idx, n, kp, mode, out_p, obs, gp, rw)
        assert torch.equal(obs, out_b) and np.array_equal(kb, kp)
        assert torch.equal(eng.pack(out_b), out_p)
        assert torch.allclose(gb, gp, rtol=1e-6, atol=1e-4)
        assert torch.equal(rw, gp.reshape(E, -1).max(dim=1).values)


@pytest.mark.parametrize("observations", ["eager", "lazy"])
@pytest.mark.parametrize("env_rng", ["shared", "independent"])
def test_vec_env_bit_plane_storage(observations, env_rng):
    """VecBreedGym(storage="bits") against the byte-layout env on the device: two horizons with
    autoreset, a population-size change, a wrapper that assigns populations, GEBV straight from bits."""
    from breedgym_amd import PackedPopulation, VecBreedGym
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
    from breedgym_amd.vector import SelectionScores

    gm = synthetic_genetic_map(1300, n_chr=4, n_traits=2, seed=3)
    geno = synthetic_genotypes(50, 1300, seed=4)
    kw = dict(num_envs=5, initial_population=geno, genetic_map=gm, individual_per_gen=24, num_generations=4,
              reward_shaping=True, env_rng=env_rng)
    ref = VecBreedGym(storage="bytes", **kw)
    env = VecBreedGym(storage="bits", observations=observations, **kw)
    p0, i0 = ref.reset(seed=5)
    p1, i1 = env.reset(seed=5)
    assert isinstance(p1, PackedPopulation) == (observations == "lazy")
    assert np.array_equal(np.asarray(p1), np.asarray(p0)) and torch.equal(i0["GEBV"], i1["GEBV"])
    rng = np.random.default_rng(1)
    for step in range(9):
        a = rng.integers(0, 24, (5, 24 if step != 5 else 17, 2)).astype(np.int32)
        o0, r0, _, t0, f0 = ref.step(a)
        o1, r1, _, t1, f1 = env.step(a)
        assert np.array_equal(np.asarray(o1), np.asarray(o0))
        assert np.array_equal(r0, r1) and np.array_equal(t0, t1)
        assert torch.allclose(f0["GEBV"].as_subclass(torch.Tensor), f1["GEBV"].as_subclass(torch.Tensor), rtol=1e-6, atol=1e-4)
        assert torch.allclose(env.get_info()["GEBV"].as_subclass(torch.Tensor),
                              ref.get_info()["GEBV"].as_subclass(torch.Tensor), rtol=1e-6, atol=1e-4)
        assert np.array_equal(env.simulator.random_key, ref.simulator.random_key)
    w0, w1 = SelectionScores(ref, k=6), SelectionScores(env, k=6)
    w0.reset(seed=2)
    w1.reset(seed=2)
    for _ in range(3):
        s = ref.get_info()["GEBV"][..., 0]
        o0 = w0.step(s)[0]
        o1 = w1.step(s)[0]
        assert np.array_equal(np.asarray(o1), np.asarray(o0))


@pytest.mark.parametrize("observations", ["eager", "lazy"])
def test_block_pool_equals_fresh_blocks(observations, rng):
    """The default pool of reusable output blocks (bh_step) against a fresh block per step (block_pool=0,
    bh_cross_packed): same observations, GEBV, rewards and key chains over two episodes incl. the autoreset; whatever
    the caller keeps -- every observation, GEBV and device-side reward of the run here -- stays what it was (a block
    is reused only when nothing handed out of it is referenced); a loop that keeps nothing alternates two blocks."""
    from breedgym_amd import VecBreedGym
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    E, n, N, M, G = 3, 40, 50, 777, 4
    gm = synthetic_genetic_map(M, n_chr=3, seed=2)
    geno = synthetic_genotypes(N, M, seed=3)
    kw = dict(num_envs=E, initial_population=geno, individual_per_gen=n, genetic_map=gm, num_generations=G,
              reward_shaping=True, sync_rewards=False, observations=observations)
    pool, fresh = VecBreedGym(**kw), VecBreedGym(block_pool=0, **kw)
    assert pool.reset(seed=9)[0].shape == fresh.reset(seed=9)[0].shape
    assert pool._ring is not None and fresh._ring is None
    held = []  # everything handed out, with a host copy made at once
    for t in range(2 * G + 1):
        a = rng.integers(0, n, (E, n, 2), dtype=np.int32)
        o1, r1, _, tr1, i1 = pool.step(a)
        o2, r2, _, tr2, i2 = fresh.step(a)
        assert np.array_equal(np.asarray(o1), np.asarray(o2)), t
        assert np.array_equal(np.asarray(i1["GEBV"]), np.asarray(i2["GEBV"]))
        assert np.array_equal(np.asarray(r1), np.asarray(r2)) and np.array_equal(tr1, tr2)
        assert np.array_equal(pool.simulator.random_key, fresh.simulator.random_key)
        assert np.array_equal(pool.random_key, fresh.random_key)
        assert np.array_equal(np.asarray(pool.populations), np.asarray(fresh.populations))
        # kept in different ways: the object itself, a view of it, only its GEBV
        held.append((o1 if t % 3 == 0 else (o1[1] if t % 3 == 1 else None), i1["GEBV"][..., 0], r1,
                     np.asarray(o1).copy(), np.asarray(i1["GEBV"]).copy(), np.asarray(r1).copy()))
        for obs, g, r, obs0, g0, r0 in held:
            if obs is not None:
                assert np.array_equal(np.asarray(obs), obs0 if obs.shape == obs0.shape else obs0[1]), t
            assert np.array_equal(np.asarray(g), g0[..., 0]) and np.array_equal(np.asarray(r), r0), t
    assert len(pool._ring.slots) > 2  # the pool grew to keep what is held
    del held, o1, i1, r1
    before = len(pool._ring.slots)
    used = set()
    for t in range(12):  # nothing kept: two blocks in turn (the rest stay idle)
        pool.step(rng.integers(0, n, (E, n, 2), dtype=np.int32))
        used.add(pool._prows.data_ptr())
    assert len(pool._ring.slots) == before and len(used) <= 3, (before, len(used))
    # holders that are not torch views: a DLPack capsule, a DLPack consumer, an object exporting
    # __cuda_array_interface__ that keeps its tensor (what CuPy / Numba arrays do) -- the slot stays out of use
    if observations == "eager":
        class Exported:  # a consumer of __cuda_array_interface__: the pointer AND the exporting object
            def __init__(self, t):
                self.owner, self.iface = t, t.as_subclass(torch.Tensor).__cuda_array_interface__

        holds = (("capsule", lambda o, g: torch.utils.dlpack.to_dlpack(o.as_subclass(torch.Tensor).view(torch.uint8))),
                 ("consumer", lambda o, g: torch.from_dlpack(g.as_subclass(torch.Tensor))),
                 ("interface", lambda o, g: Exported(g)))
        for kind, hold in holds:
            o, _, _, _, i = pool.step(rng.integers(0, n, (E, n, 2), dtype=np.int32))
            o0, g0 = np.asarray(o).view(np.uint8).copy(), np.asarray(i["GEBV"]).copy()
            ptr_o = o.data_ptr()
            h = hold(o, i["GEBV"])
            del o, i
            for _ in range(6):
                pool.step(rng.integers(0, n, (E, n, 2), dtype=np.int32))
                if kind == "capsule":  # (at a horizon step the observation and the GEBV come from different slots)
                    assert pool._populations.data_ptr() != ptr_o, kind  # the held slot is not written again
            if kind == "capsule":
                assert np.array_equal(torch.from_dlpack(h).cpu().numpy(), o0)
            else:
                assert np.array_equal((h if kind == "consumer" else h.owner).as_subclass(torch.Tensor).cpu().numpy(), g0), kind
            del h
    # a wrapper may replace the populations (breeding_programs_env.py:56): the next step reads them
    fresh.reset(seed=4)
    pool.reset(seed=4)
    newpop = np.asarray(fresh.populations).copy()[:, ::-1]
    pool.populations, fresh.populations = newpop, newpop
    a = rng.integers(0, n, (E, n, 2), dtype=np.int32)
    assert np.array_equal(np.asarray(pool.step(a)[0]), np.asarray(fresh.step(a)[0]))
    # a pool that may not grow falls back to fresh blocks when everything is held
    tiny = VecBreedGym(block_pool=2, **kw)
    tiny.reset(seed=9)
    keep = [tiny.step(rng.integers(0, n, (E, n, 2), dtype=np.int32))[0] for _ in range(5)]
    assert len({k.data_ptr() if hasattr(k, "data_ptr") else id(k) for k in keep}) == 5 and len(tiny._ring.slots) == 2
    with pytest.raises(ValueError):
        VecBreedGym(block_pool=1, **kw)


@pytest.mark.parametrize("storage,observations", [("bits", "eager"), ("bits", "lazy"), ("bytes", "eager")])
def test_checkpoint_resume_on_device(storage, observations, rng):
    """get_state / set_state (the env state of vec_env.py:70-73): an episode continued in a fresh env on the device is
    the episode continued in the old one, through the horizon and its autoreset"""
    from breedgym_amd import VecBreedGym
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    E, n, M = 3, 33, 1500
    gm = synthetic_genetic_map(M, n_chr=3, seed=2)
    geno = synthetic_genotypes(60, M, seed=3)
    kw = dict(num_envs=E, initial_population=geno, individual_per_gen=n, genetic_map=gm, num_generations=5,
              reward_shaping=True, storage=storage, observations=observations)
    a, b = VecBreedGym(**kw), VecBreedGym(**kw)
    a.reset(seed=5)
    for _ in range(2):
        a.step(rng.integers(0, n, (E, n, 2), dtype=np.int32))
    b.reset(seed=99)
    b.set_state(a.get_state())
    for _ in range(5):
        act = rng.integers(0, n, (E, n, 2), dtype=np.int32)
        pa, ra, _, ta, ia = a.step(act)
        pb, rb, _, tb, ib = b.step(act)
        assert np.array_equal(np.asarray(pa), np.asarray(pb)) and np.array_equal(ra, rb) and np.array_equal(ta, tb)
        assert np.array_equal(np.asarray(ia["GEBV"]), np.asarray(ib["GEBV"]))
    assert np.array_equal(a.random_key, b.random_key) and np.array_equal(a.simulator.random_key, b.simulator.random_key)


@pytest.mark.parametrize("E,NP,NO,M,T", [
    (24, 64, 256, 10_000, 1),   # 6144 rows x 20 KB = 123 MB of observation: beyond the Infinity-Cache policy -> non-temporal
                                # stores, digits from memory (cross.hip, obs_store_nt)
    (8, 300, 1024, 1200, 1),    # 1024 keyed crosses (more than 512) shared by 8 envs: four-wave mask groups and the pipelined
                                # stream, not the one-wave form (mask_waves_for_runahead); 8192 rows: two rows per wave
    (8, 300, 1024, 1200, 2),    # ... with two traits (one vector per lane and trip)
    (2, 40, 700, 3001, 1),      # 700 keyed crosses for 1400 rows: more than one per row workgroup -> one-wave mask groups
])
def test_packed_cross_large_batches_vs_oracle(E, NP, NO, M, T, rng):
    """The shapes on either side of this round's dispatch rules, two generations each (the second reads the masks the
    first launch made ahead), every output against the oracle."""
    r = (rng.random(M) * 0.03).astype(np.float32)
    r[:: max(1, M // 7)] = 0.5
    eff = rng.laplace(0, 3, (M, T)).astype(np.float32)
    eng = make_engine(r, eff)
    co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    pool = rng.integers(0, 2, (E, NP, M, 2), dtype=np.uint8)
    seed = int(rng.integers(0, 2**32))
    key_ref = co.key_from_seed(seed)
    key = np.array(key_ref, np.uint32).copy()
    prow = eng.pack(eng.upload_population(pool.reshape(E * NP, M, 2)))
    n_pool = NP
    for gen in range(2):
        idx = rng.integers(0, n_pool, (E, NO, 2), dtype=np.int32)
        outp, obs = eng.empty_packed(E * NO), eng.empty_population(E * NO)
        g, rew = eng.empty((E, NO, T), torch.float32), eng.empty_reward(E)
        eng.cross_packed(prow, E, n_pool, torch.from_numpy(idx).cuda(), NO, key, 0, outp, obs, g, rew)
        ref, key_ref = co.cross(pool, idx, r, key_ref, 0)
        got = eng.as_genotypes(obs, E, NO).cpu().numpy().view(np.uint8).reshape(ref.shape)
        assert np.array_equal(got, ref), f"generation {gen}"
        assert np.array_equal(packed_host(outp), pack_numpy(ref.reshape(E * NO, M, 2)))
        assert np.array_equal(key, key_ref)
        g64 = co.gebv(ref, eff, want64=True)[1]
        assert gebv_close(g.cpu().numpy(), g64)
        assert np.allclose(rew.cpu().numpy(), g64.max(axis=(1, 2)), rtol=GEBV_RTOL)
        pool, prow, n_pool = ref, outp, NO
