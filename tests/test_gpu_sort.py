"""GPU: the draws and rankings that no longer fit one workgroup's LDS -- JAX's sort-based shuffle by the wanted-only
route (csrc/shuffle_sparse.h: the wanted front propagated backwards through the rounds, nothing sorted) or, where
nearly all of a long list is wanted, as radix sorts in HBM (csrc/sort_kernels.h) -- against the CPU oracle, bit-exact, through the C ABI: the founder draw of
`VecBreedGym.reset` (breedgym/vector/vec_env.py:22-27) on large germplasm, `SelectionScores._convert_actions`
(breedgym/vector/vec_wrappers.py:61-79) at the shapes BASELINE's configs[2] and configs[3] give it with the
reference's default k = n // 10, and `lax.top_k` with more than 16384 winners."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["original", "partitionable"])
def layout(request):
    co.set_threefry_layout(request.param)
    yield request.param
    co.set_threefry_layout("original")


def tiny_engine(layout):
    from breedgym_amd import HipEngine

    return HipEngine(np.full(8, 0.1, np.float32), np.ones((8, 1), np.float32), threefry_layout=layout)


@pytest.mark.parametrize("N,n,E", [(8193, 50, 2), (10000, 10000, 3), (16385, 4096, 2), (65536, 8192, 2),
                                     (65536, 65536, 1), (100001, 1000, 5), (1 << 20, 4096, 1),
                                     ((1 << 20) + 3, 1000, 3), (262144, 262144, 1), (262145, 16384, 2),
                                     (300000, 100000, 1), (400001, 400001, 1)])  # the last two: the radix sort (most of a long list)
def test_founder_draw_large_germplasm(N, n, E, layout):
    eng = tiny_engine(layout)
    key0 = co.key_from_seed(1234)
    got, key = eng.founders(key0, E, N, n)
    ref, refkey = co.founders(key0, E, N, n)
    assert np.array_equal(got.cpu().numpy().reshape(E, n), ref), (N, n, E)
    assert np.array_equal(key, refkey)


def test_founder_draw_through_the_env_window(layout):
    """envs [2, 5) of 7 draw what envs 2..4 of the whole would (bh_set_env_window), on the radix-sort route"""
    eng = tiny_engine(layout)
    N, n = 20000, 777
    key0 = co.key_from_seed(99)
    ref, refkey = co.founders(key0, 7, N, n)
    eng.set_env_window(2, 7)
    got, key = eng.founders(key0, 3, N, n)
    assert np.array_equal(got.cpu().numpy().reshape(3, n), ref[2:5])
    assert np.array_equal(key, refkey)


@pytest.mark.parametrize("E,n,k,nc", [(3, 300, 200, 300),        # 19,900 pairs: two rounds, three tiles
                                        (4, 2048, 204, 2048),      # configs[2] with the reference's default k
                                        (2, 9000, 40, 780),        # n beyond the LDS kernel, pairs inside it
                                        (2, 20000, 1000, 5000),    # top-k by radix select, 499,500 pairs
                                        (2, 20000, 800, 20000),    # 319,600 pairs, 20,000 of them wanted: the radix sort
                                        (1, 65536, 6553, 8192)])   # configs[3]: 21.5 M pairs, three rounds
def test_selection_actions_large(E, n, k, nc, layout, rng):
    eng = tiny_engine(layout)
    s = rng.standard_normal((E, n)).astype(np.float32)
    s[:, 5] = s[:, 11]  # a tie inside or at the edge of the top-k
    s[:, n // 2] = s[:, 0]
    key0 = co.key_from_seed(31)
    got, key = eng.selection_actions(torch.from_numpy(s), k, nc, key0)
    ref, refkey = co.selection_actions(s, k, nc, key0)
    assert np.array_equal(got.cpu().numpy(), ref), (E, n, k, nc)
    assert np.array_equal(key, refkey)


@pytest.mark.parametrize("E,n", [(5, 785), (3, 2000), (6, 130)])
def test_selection_actions_of_one_pair_many_times(E, n, layout, rng):
    """k = 2: ONE diallel pair, JAX's shuffle of one element has zero rounds, and every slot of the generation takes
    that pair.  `shuffle_lds` returned the identity it had just written without a barrier: the waves behind the first
    read perm[0] before it was there -- rows [64, blockDim) of an env crossed from whatever LDS held, about once in a
    few hundred calls (profiles/r05_fuzz_env_mismatch.txt, found by tools/fuzz_env.py).  Many calls, dirty LDS in
    between (a founder draw of another size), every one against the oracle."""
    eng = tiny_engine(layout)
    key = co.key_from_seed(77)
    for it in range(300):
        s = np.round(rng.normal(0, 2, (E, n)), 1).astype(np.float32)  # scores with ties
        ref, refkey = co.selection_actions(s, 2, 1, key)
        got, key2 = eng.selection_actions(torch.from_numpy(s), 2, 1, key)
        assert np.array_equal(got.cpu().numpy(), ref), (E, n, it)
        assert np.array_equal(key2, refkey)
        key = refkey
        if it % 3 == 0:
            eng.founders(key, 2, 500 + it, 200)  # something else through the same CUs' LDS


def test_selection_actions_large_through_the_env_window(rng):
    eng = tiny_engine("original")
    E, n, k, nc = 5, 2048, 204, 1024
    s = rng.standard_normal((E, n)).astype(np.float32)
    key0 = co.key_from_seed(8)
    ref, refkey = co.selection_actions(s, k, nc, key0)
    eng.set_env_window(1, E)
    got, key = eng.selection_actions(torch.from_numpy(s[1:4]), k, nc, key0)
    assert np.array_equal(got.cpu().numpy(), ref[1:4])
    assert np.array_equal(key, refkey)


@pytest.mark.parametrize("rows,n,k", [(1, 100000, 20000), (3, 40000, 16385), (2, 65536, 65536)])
def test_topk_more_winners_than_lds(rows, n, k, rng):
    eng = tiny_engine("original")
    s = rng.standard_normal((rows, n)).astype(np.float32)
    s[:, ::7] = np.round(s[:, ::7], 1)  # many exact ties
    s[0, 3], s[0, 4] = np.inf, -np.inf
    got = eng.topk(torch.from_numpy(s), k).cpu().numpy()
    for r in range(rows):
        assert np.array_equal(got[r], co.topk(s[r], k))


def test_selection_scores_env_with_reference_defaults_at_c3():
    """SelectionScores(VecBreedGym(configs[2])) with the reference's defaults: k = n // 10 = 204 gives 20,706 pairs --
    more than the generation's 2048 slots, a ValueError like the reference's (vec_wrappers.py:46-51); with
    n_crosses = n the wrapper steps, and its low-level actions are the oracle's."""
    from breedgym_amd import VecBreedGym
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
    from breedgym_amd.vector import SelectionScores

    E, n, N, M = 4, 2048, 2048, 100_000
    gm = synthetic_genetic_map(M, n_chr=21, seed=3)
    geno = synthetic_genotypes(N, M, seed=4)
    base = VecBreedGym(num_envs=E, initial_population=geno, individual_per_gen=n, genetic_map=gm, num_generations=3)
    with pytest.raises(ValueError):
        SelectionScores(base)
    env = SelectionScores(base, n_crosses=n)
    assert env.k == 204
    _, infos = env.reset(seed=5)
    _, env_key = co.founders(co.key_from_seed(5), E, N, n)
    for _ in range(2):  # SelectionScores.step, taken apart to see the low-level actions
        scores = infos["GEBV"][..., 0]
        ref_low, env_key = co.selection_actions(np.asarray(scores), env.k, n, env_key)
        low = env._convert_actions(scores)
        assert np.array_equal(low.cpu().numpy(), ref_low)
        assert np.array_equal(np.asarray(env.random_key), env_key)
        _, _, _, _, infos = env.env.step(low)
    obs, rew, ter, tru, infos = env.step(infos["GEBV"][..., 0])  # ... and whole: the horizon step
    assert tru.all() and obs.shape == (E, n, M, 2)


def test_release_scratch_then_draw_again(layout):
    """bh_release_scratch gives the sort scratch back; the next large draw allocates it again and is the same draw."""
    eng = tiny_engine(layout)
    key0 = co.key_from_seed(77)
    a, _ = eng.founders(key0, 2, 70000, 3000)
    eng.release_scratch()
    b, _ = eng.founders(key0, 2, 70000, 3000)
    ref, _ = co.founders(key0, 2, 70000, 3000)
    assert np.array_equal(a.cpu().numpy().reshape(2, 3000), ref) and np.array_equal(b.cpu().numpy().reshape(2, 3000), ref)
    s = np.random.default_rng(1).standard_normal((2, 9000)).astype(np.float32)
    got, _ = eng.selection_actions(torch.from_numpy(s), 300, 2000, key0)
    eng.release_scratch()
    ref_a, _ = co.selection_actions(s, 300, 2000, key0)
    assert np.array_equal(got.cpu().numpy(), ref_a)


@pytest.mark.parametrize("n,k", [(4097, 1), (4097, 2048), (5000, 300), (8192, 25), (8192, 819), (8192, 4096), (12345, 1234),
                                 (16384, 1638), (16384, 8192)])
def test_topk_mid_rows_by_radix_select(n, k, rng):
    """Rows of 4097 .. 16384 scores with at most half wanted go through the radix select + a sort of the winners (round 5;
    before: one workgroup's sort of the whole row): jax.lax.top_k's order -- ties to the lower index, -0 below +0,
    infinities -- on scores that are MOSTLY ties, several rows at once."""
    import torch
    from breedgym_amd import HipEngine

    eng = HipEngine(np.full(64, 0.01, np.float32), np.ones((64, 1), np.float32))
    s = np.round(rng.normal(0, 2, (3, n))).astype(np.float32)  # a handful of distinct values
    s[0, rng.integers(0, n, 40)] = np.inf
    s[1, rng.integers(0, n, 40)] = -np.inf
    s[2, rng.integers(0, n, n // 3)] = -0.0
    s[2, rng.integers(0, n, n // 3)] = 0.0
    got = eng.topk(torch.from_numpy(s).to(eng.device), k).cpu().numpy()
    for e in range(3):
        assert np.array_equal(got[e], co.topk(s[e], k)), (n, k, e)
