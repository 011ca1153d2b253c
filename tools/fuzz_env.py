"""Randomised VecBreedGym episodes against the CPU oracle's chain, for as long as the clock allows -- aimed at the host
logic around the step (breedgym_amd/vector/vec_env.py): the ring of output blocks, founders drawn a step or an episode
ahead, the reset gathered by the horizon step's launch, and everything that voids what was made ahead.

Each case: random (envs, individuals per generation, germplasm, markers, traits, episode length), storage
(bits / bytes), observations (eager / lazy), key mode, reward shaping, autoreset; then ~40 events, each a step with
random actions or -- now and then -- one of: the env key advanced by hand (what SelectionScores does per step), a
reset(seed=...), a reset() in the middle of an episode, a reset(options={"individual_per_gen": n'}), a get_state /
set_state round trip into a FRESH env, an observation kept by the caller across many steps (its block must not be
reused).  Checked at every step: the observation (when handed out), GEBV, reward, truncation flags, both key chains.
usage: fuzz_env.py [seconds] [seed]      FUZZ_ENV_FIXED="E,n,N,M,T,G,k,n_crosses" pins every case to one shape"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np

from breedgym_amd import VecBreedGym, rng as hrng
from breedgym_amd.vector import SelectionScores
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
from oracle import c_oracle as co

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DIAG = {}
FIXED = [int(x) for x in __import__("os").environ.get("FUZZ_ENV_FIXED", "").split(",") if x]
co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))


def u8(x, shape):
    return np.asarray(x).view(np.uint8).reshape(shape)


def diagnose(env, pop, ref, got, g64, gm, tag):
    """What a GEBV mismatch looked like: where, by how much, whether the offspring themselves are right, whether the
    matrix-core passes' cells were left dirty, and what a second evaluation of the same rows gives."""
    import ctypes

    from breedgym_amd import _lib

    bad = np.argwhere(np.abs(got - g64) > 1e-5 * np.maximum(np.abs(g64), 1.0))
    print("MISMATCH", tag, flush=True)
    print(" GEBV shape", got.shape, "bad cells", len(bad), "first", bad[:8].tolist(), flush=True)
    for e, i, t in bad[:8]:
        print(f"  [{e},{i},{t}] got {got[e, i, t]!r} want {g64[e, i, t]!r} diff {got[e, i, t] - g64[e, i, t]!r}", flush=True)
    obs = u8(pop, ref.shape)
    print(" offspring equal to the oracle's:", bool(np.array_equal(obs, ref)), flush=True)
    try:
        st = u8(env.populations, ref.shape)
        print(" the env's state equal to the observation:", bool(np.array_equal(st, obs)), " to the oracle's:", bool(np.array_equal(st, ref)), flush=True)
    except Exception as ex:
        print(" state not comparable:", repr(ex), flush=True)
    par, acts = DIAG.get("parents"), DIAG.get("actions")
    for e in range(ref.shape[0]):
        wrong = np.flatnonzero((obs[e] != ref[e]).reshape(ref.shape[1], -1).any(axis=1))
        if not len(wrong):
            continue
        r = int(wrong[0])
        d = np.argwhere(obs[e, r] != ref[e, r])
        runs = np.flatnonzero(np.diff(wrong) > 1)
        print(f" env {e}: {len(wrong)} wrong rows, first {r} last {int(wrong[-1])}, {len(runs) + 1} run(s); row {r}: {len(d)} alleles differ, "
              f"markers {int(d[0, 0])} .. {int(d[-1, 0])}", flush=True)
        same = [int(x) for x in np.flatnonzero((ref[e] == obs[e, r]).reshape(ref.shape[1], -1).all(axis=1))[:4]]
        print(f"   row {r} as handed out equals the oracle's row(s) {same} of the env", flush=True)
        if par is not None and acts is not None:
            a0, a1 = (int(x) for x in acts[e, r])
            npar = par.shape[1]
            c0, c1 = min(max(a0 + npar if a0 < 0 else a0, 0), npar - 1), min(max(a1 + npar if a1 < 0 else a1, 0), npar - 1)
            p0, p1 = par[e, c0], par[e, c1]
            v0 = int(((obs[e, r, :, 0] != p0[:, 0]) & (obs[e, r, :, 0] != p0[:, 1])).sum())
            v1 = int(((obs[e, r, :, 1] != p1[:, 0]) & (obs[e, r, :, 1] != p1[:, 1])).sum())
            print(f"   parents ({a0}, {a1}) of {npar}: alleles that neither haplotype of the parent has: gamete 0 {v0}, gamete 1 {v1}; "
                  f"actions of the env all equal: {bool((acts[e] == acts[e, 0]).all())}", flush=True)
    eng = env.simulator.engine
    dirty = ctypes.c_int64(-1)
    _lib.check(eng._L.bh_debug_count_dirty_cells(eng._h, ctypes.byref(dirty)))
    print(" dirty cells of the matrix-core passes:", dirty.value, flush=True)
    try:
        rows = eng.upload_population(np.ascontiguousarray(ref.reshape(-1, ref.shape[-2], 2)))
        again = eng.gebv(rows).cpu().numpy().astype(np.float64).reshape(g64.shape)
        print(" bh_gebv of the oracle's rows: max |diff|", float(np.abs(again - g64).max()), flush=True)
        g2 = eng.empty((rows.shape[0], g64.shape[-1]), __import__("torch").float32)
        eng.gather_packed(eng.pack(rows), None, rows.shape[0], None, None, g2)
        print(" gather_packed GEBV of the same rows: max |diff|",
              float(np.abs(g2.cpu().numpy().astype(np.float64).reshape(g64.shape) - g64).max()), flush=True)
    except Exception as ex:  # the diagnosis must not hide the mismatch
        print(" second evaluation failed:", repr(ex), flush=True)


def one_case(case):
    E = int(rng.integers(1, 7))
    n = int(rng.choice([rng.integers(1, 20), rng.integers(20, 300), rng.integers(300, 1200)], p=[0.3, 0.5, 0.2]))
    N = int(rng.integers(max(n, 2), max(n, 2) + 60))
    M = int(rng.choice([rng.integers(8, 200), rng.integers(200, 3000), rng.integers(3000, 12000)], p=[0.4, 0.4, 0.2]))
    T = int(rng.choice([1, 1, 2, 3]))
    G = int(rng.integers(1, 6))
    storage = "bits" if rng.random() < 0.75 else "bytes"
    observations = "lazy" if storage == "bits" and rng.random() < 0.3 else "eager"
    env_rng = "shared" if rng.random() < 0.7 else "independent"
    shaping, autoreset = bool(rng.random() < 0.6), bool(rng.random() < 0.85)
    if FIXED:  # FUZZ_ENV_FIXED="E,n,N,M,T,G,k,n_crosses": every case at one shape (bit planes, eager, shared keys)
        E, n, N, M, T, G = FIXED[:6]
        storage, observations, env_rng = "bits", "eager", "shared"
    gm = synthetic_genetic_map(M, n_chr=int(rng.integers(1, 5)), n_traits=T, seed=int(rng.integers(0, 1000)))
    geno = synthetic_genotypes(N, M, seed=int(rng.integers(0, 1000)))
    kw = dict(num_envs=E, initial_population=geno, individual_per_gen=n, num_generations=G, autoreset=autoreset,
              reward_shaping=shaping, env_rng=env_rng, storage=storage, observations=observations, genetic_map=gm)
    what = f"case {case}: E={E} n={n} N={N} M={M} T={T} G={G} {storage}/{observations} {env_rng} shaping={shaping} autoreset={autoreset}"
    env = VecBreedGym(**kw)
    # a third of the cases step through SelectionScores (vec_wrappers.py:15-86): scores in, the k best paired at random,
    # the env key advanced by its shuffle every step -- which voids everything drawn ahead, every step
    wrap_k = int(rng.integers(2, min(n, 12) + 1)) if n >= 3 and rng.random() < 0.33 else 0
    wrap_nc = int(rng.integers(1, min(wrap_k * (wrap_k - 1) // 2, n) + 1)) if wrap_k else 0
    if FIXED:
        wrap_k, wrap_nc = FIXED[6:8]
    stepper = SelectionScores(env, k=wrap_k, n_crosses=wrap_nc) if wrap_k else env
    what += f" SelectionScores(k={wrap_k}, n_crosses={wrap_nc})" if wrap_k else ""
    mode = 0 if env_rng == "shared" else 1
    seed = int(rng.integers(0, 2**31))
    pop, info = env.reset(seed=seed)
    f, env_key = co.founders(co.key_from_seed(seed), E, N, n)
    ref, sim_key, step_idx = geno.view(np.uint8)[f], co.key_from_seed(seed), 0
    assert np.array_equal(u8(pop, ref.shape), ref), what + ": first reset"
    kept = []  # (observation the caller kept, what it must still be)

    def check_reset(pop, info, n_now, tag):
        assert np.array_equal(u8(pop, ref.shape), ref), what + f": {tag}: observation"
        assert np.allclose(np.asarray(info["GEBV"]), co.gebv(ref, gm.marker_effects), rtol=1e-5, atol=1e-5), what + f": {tag}: GEBV"
        assert np.array_equal(env.random_key, env_key), what + f": {tag}: env key"

    for ev in range(40):
        roll = rng.random()
        if step_idx == G and not autoreset:
            roll = 0.95  # the episode is over and nothing resets it: reset by hand
        if roll < 0.72:
            if wrap_k:
                sc = np.round(rng.normal(0, 2, (E, n)), int(rng.integers(0, 3))).astype(np.float32)  # scores with ties
                key_before = env_key
                a, env_key = co.selection_actions(sc, wrap_k, wrap_nc, env_key)
                # the actions themselves, not only what comes of them (the same conversion the wrapper is about to make)
                got_a, _ = env.simulator.engine.selection_actions(sc, wrap_k, wrap_nc, key_before)
                assert np.array_equal(got_a.cpu().numpy(), a), what + f": event {ev}: SelectionScores' actions"
                pop, rews, ter, tru, info = stepper.step(sc)
            else:
                a = rng.integers(0, n, (E, n, 2), dtype=np.int32)
                pop, rews, ter, tru, info = env.step(a)
            DIAG["parents"], DIAG["actions"] = ref, np.asarray(a)
            ref, sim_key = co.cross(ref, a, gm.recombination_vec, sim_key, mode)
            step_idx += 1
            done = step_idx == G
            g64 = co.gebv(ref, gm.marker_effects, want64=True)[1]
            tag = what + f": event {ev} (step {step_idx} of {G})"
            got = np.asarray(info["GEBV"]).astype(np.float64)
            if not np.all(np.abs(got - g64) <= 1e-5 * np.maximum(np.abs(g64), 1.0)):
                diagnose(env, pop, ref, got, g64, gm, tag)
            assert np.all(np.abs(got - g64) <= 1e-5 * np.maximum(np.abs(g64), 1.0)), tag + ": GEBV"
            want = g64.max(axis=(1, 2)) if (shaping or done) else np.zeros(E)
            assert np.allclose(np.asarray(rews, dtype=np.float64), want, rtol=1e-5, atol=1e-6), tag + ": reward"
            assert np.array_equal(np.asarray(tru), np.full(E, done)) and not np.asarray(ter).any(), tag + ": flags"
            assert np.array_equal(env.simulator.random_key, sim_key), tag + ": simulator key"
            if done and autoreset:
                f, env_key = co.founders(env_key, E, N, n)
                ref, step_idx = geno.view(np.uint8)[f], 0
            assert np.array_equal(u8(pop, ref.shape), ref), tag + ": observation"
            assert np.array_equal(env.random_key, env_key), tag + ": env key"
            if rng.random() < 0.15 and len(kept) < 3:
                kept.append((pop, ref.copy()))
        elif roll < 0.80:  # somebody advances the env key (SelectionScores' shuffle does, every step)
            env.random_key = hrng.split(env.random_key, 2, env.simulator.threefry_layout)[0]
            env_key = co.split(env_key, 2)[0]
        elif roll < 0.86:  # reset(seed)
            seed = int(rng.integers(0, 2**31))
            pop, info = env.reset(seed=seed)
            f, env_key = co.founders(co.key_from_seed(seed), E, N, n)
            ref, sim_key, step_idx = geno.view(np.uint8)[f], co.key_from_seed(seed), 0
            check_reset(pop, info, n, f"event {ev}: reset(seed)")
        elif roll < 0.91 and not wrap_k:  # a population size of another kind (vec_env.py:118-119)
            n = int(rng.integers(1, min(N, 400) + 1))
            pop, info = env.reset(options={"individual_per_gen": n})
            f, env_key = co.founders(env_key, E, N, n)
            ref, step_idx = geno.view(np.uint8)[f], 0
            check_reset(pop, info, n, f"event {ev}: reset(options n={n})")
        elif roll < 0.96 or not hasattr(env, "get_state"):  # plain reset, wherever the episode is
            pop, info = env.reset()
            f, env_key = co.founders(env_key, E, N, n)
            ref, step_idx = geno.view(np.uint8)[f], 0
            check_reset(pop, info, n, f"event {ev}: reset()")
        else:  # checkpoint into a fresh env
            state = env.get_state()
            env = VecBreedGym(**{**kw, "individual_per_gen": n})
            env.set_state(state)
            stepper = SelectionScores(env, k=wrap_k, n_crosses=wrap_nc) if wrap_k else env
            assert np.array_equal(u8(env.populations, ref.shape), ref), what + f": event {ev}: set_state"
        for obs, must in kept:
            assert np.array_equal(u8(obs, must.shape), must), what + f": event {ev}: an observation the caller kept was overwritten"
    return what


t0, case, last = time.time(), 0, ""
while time.time() - t0 < budget:
    last = one_case(case)
    case += 1
    if case % 10 == 0:
        print(f"{case} cases, {time.time() - t0:.0f} s; last: {last}", flush=True)
print(f"fuzz_env: {case} cases x 40 events in {time.time() - t0:.0f} s, 0 mismatches")
