"""NumPy restatement of the reference's op graph (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Deliberately *unfused*, array-at-a-time, like the reference's CPU execution through
chromax/JAX: materialised parent gather (``breedgym/vector/vec_env.py:89-90``),
``uniform(M) < r``, cumulative XOR, ``take_along_axis``, ``sum(-1) @ effects``
(``scripts/optimize_diff_cross.py:16-17``).  It is the readable specification, the
second opinion for the C oracle, and the "B1" CPU baseline of BASELINE.md.

PARITY UNPINNED: chromax/jax are not installed; the key plumbing of
``chromax.Simulator.cross`` is restated from its published source from memory and is
marked [EXTERNAL/UNVERIFIED].
"""
import math

import numpy as np

_R = (13, 15, 26, 6, 17, 29, 16, 24)
_U32 = np.uint32


def _rotl(x, r):
    return (x << _U32(r)) | (x >> _U32(32 - r))


def threefry2x32(key, c0, c1):
    """Threefry-2x32-20 on uint32 arrays c0, c1 (same shape) with one key pair."""
    with np.errstate(over="ignore"):
        k0, k1 = _U32(key[0]), _U32(key[1])
        ks = (k0, k1, k0 ^ k1 ^ _U32(0x1BD11BDA))
        x0 = np.asarray(c0, _U32) + ks[0]
        x1 = np.asarray(c1, _U32) + ks[1]
        for blk in range(5):
            rot = _R[4:] if blk & 1 else _R[:4]
            for r in rot:
                x0 = x0 + x1
                x1 = _rotl(x1, r)
                x1 = x1 ^ x0
            x0 = x0 + ks[(blk + 1) % 3]
            x1 = x1 + ks[(blk + 2) % 3] + _U32(blk + 1)
    return x0, x1


def key_from_seed(seed):
    return np.array([0, int(seed) & 0xFFFFFFFF], _U32)


def _threefry_iota(key, n):
    """jax `threefry_2x32(key, iota(n))`: pad to even with a 0 counter, hash the two halves."""
    cnt = np.arange(n, dtype=_U32)
    if n % 2:
        cnt = np.concatenate([cnt, np.zeros(1, _U32)])
    h = len(cnt) // 2
    o0, o1 = threefry2x32(key, cnt[:h], cnt[h:])
    return np.concatenate([o0, o1])[:n]


LAYOUT = "original"  # or "partitionable" (the default of jax >= 0.5); set_threefry_layout() switches


def set_threefry_layout(layout):
    global LAYOUT
    assert layout in ("original", "partitionable")
    LAYOUT = layout


def random_bits(key, n):
    if LAYOUT == "partitionable":  # jax/_src/prng.py _threefry_random_bits_partitionable: counts (0, i), hi ^ lo
        o0, o1 = threefry2x32(key, np.zeros(n, _U32), np.arange(n, dtype=_U32))
        return o0 ^ o1
    return _threefry_iota(key, n)


def split(key, num):
    if LAYOUT == "partitionable":  # _threefry_split_foldlike: key k = both words of TF(key, (0, k))
        o0, o1 = threefry2x32(key, np.zeros(num, _U32), np.arange(num, dtype=_U32))
        return np.stack([o0, o1], axis=1)
    return _threefry_iota(key, 2 * num).reshape(num, 2)


def uniform(key, n):
    bits = random_bits(key, n)
    f = ((bits >> _U32(9)) | _U32(0x3F800000)).view(np.float32) - np.float32(1.0)
    return np.maximum(np.float32(0.0), f)


def permutation(key, N):
    """jax.random.permutation(key, N): `_shuffle` = repeated stable sort by random 32-bit keys."""
    rounds = int(math.ceil(3 * math.log(max(1, N)) / math.log(2**32 - 1)))
    x = np.arange(N, dtype=np.int32)
    key = np.asarray(key, _U32)
    for _ in range(rounds):
        key, sub = split(key, 2)
        x = x[np.argsort(random_bits(sub, N), kind="stable")]
    return x


def founders(env_key, E, N, n):
    """VecBreedGym.reset founder draw (vec_env.py:22-27,120-128)."""
    keys = split(env_key, E + 1)
    idx = np.stack([permutation(keys[1 + e], N)[:n] for e in range(E)])
    return idx, keys[0].copy()


def meiosis(individual, r, gamete_key, mutation=0.0):
    """[EXTERNAL/UNVERIFIED] chromax.functional._meiosis: (M,2) -> gamete (M,)."""
    rec_key, mut_key = split(gamete_key, 2)
    sites = uniform(rec_key, len(r)) < r
    mask = np.logical_xor.accumulate(sites).astype(np.int8)
    gamete = np.take_along_axis(individual, mask[:, None], axis=-1)[:, 0]
    if mutation > 0:  # jax.random.bernoulli(mut_key, p, (M,)) = uniform(mut_key, (M,)) < p, xor-ed into the gamete
        gamete = gamete ^ (uniform(mut_key, len(r)) < np.float32(mutation)).astype(gamete.dtype)
    return gamete, mask


def cross(parents, r, sim_key, mutation=0.0):
    """[EXTERNAL/UNVERIFIED] chromax.Simulator.cross on materialised parents (n,2,M,2).

    Returns (offspring (n,M,2), new_sim_key, masks (n,2,M))."""
    n = len(parents)
    keys = split(sim_key, 2 * n + 1)
    out = np.empty((n, parents.shape[2], 2), parents.dtype)
    masks = np.empty((n, 2, parents.shape[2]), np.uint8)
    for i in range(n):
        for p in range(2):
            out[i, :, p], masks[i, p] = meiosis(parents[i, p], r, keys[1 + 2 * i + p], mutation)
    return out, keys[0].copy(), masks


def vec_step_cross(populations, actions, r, sim_key, rng_mode=0, mutation=0.0):
    """vec_env.py:88-91: gather parents, then vmap(cross) over envs.

    rng_mode 0 reproduces vmap over an un-batched simulator key: every env gets the
    same gamete keys.  rng_mode 1 treats the E*n crosses as one flat batch."""
    E, n = actions.shape[:2]
    n_pool = populations.shape[1]
    # JAX retrieval indexing: negatives wrap once, the rest is clamped (high is inclusive,
    # vec_env.py:63-68, so index n_pool can occur)
    actions = np.clip(np.where(actions < 0, actions + n_pool, actions), 0, n_pool - 1)
    parents = populations[np.arange(E)[:, None, None], actions]  # (E,n,2,M,2) materialised
    if rng_mode == 0:
        outs, newkey = [], None
        for e in range(E):
            o, newkey, _ = cross(parents[e], r, sim_key, mutation)
            outs.append(o)
        return np.stack(outs), newkey
    flat, newkey, _ = cross(parents.reshape(E * n, 2, *parents.shape[3:]), r, sim_key, mutation)
    return flat.reshape(E, n, *flat.shape[1:]), newkey


def gebv(pop, effects):
    """sum over ploidy, dot with marker effects; f64 so it can arbitrate tolerances."""
    dosage = pop.astype(np.int8).sum(-1, dtype=np.int8)
    return dosage.astype(np.float64) @ np.asarray(effects, np.float64)


def reward_max(g):
    return np.max(g, axis=(1, 2))


def topk(scores, k):
    """jax.lax.top_k: descending in XLA's total order of the bit patterns (-0 below +0), ties to the lower index."""
    u = np.ascontiguousarray(scores, np.float32).view(np.uint32).astype(np.int64)
    order = np.where(u >> 31, 0x7FFFFFFF - u, u)  # sign-magnitude -> two's complement
    return np.argsort(-order, kind="stable")[:k].astype(np.int32)


def selection_actions(scores, k, n_crosses, env_key):
    """SelectionScores._convert_actions vmapped over envs + the key split of SelectionScores.step
    (vec_wrappers.py:61-85)."""
    scores = np.asarray(scores, np.float32)
    E, n = scores.shape
    keys = split(env_key, E + 1)
    n_off = math.ceil(n / n_crosses)
    out = np.empty((E, n, 2), np.int32)
    for e in range(E):
        best = topk(scores[e], k)
        iu = np.triu_indices(k, k=1)
        diallel = np.stack((best[iu[0]], best[iu[1]]), axis=1)
        pick = permutation(keys[1 + e], len(diallel))[:n_crosses]
        out[e] = np.repeat(diallel[pick], n_off, axis=0)[:n]
    return out, keys[0].copy()


# ---- fully vectorised variant used as the "B1" CPU baseline ------------------------------


def cross_batched(parents, r, sim_key):
    """Same result as `cross`, array-at-a-time over all crosses (the unfused reference graph)."""
    n, _, M, _ = parents.shape
    keys = split(sim_key, 2 * n + 1)
    gk = keys[1:].reshape(n * 2, 2)
    # rec_key = split(gk, 2)[0] = (TF(gk,(0,2))[0], TF(gk,(1,3))[0]) for every gamete key
    with np.errstate(over="ignore"):
        rk = np.empty_like(gk)
        for w, (c0, c1) in enumerate(((0, 2), (1, 3))):
            rk[:, w] = _tf_multi(gk, np.full(len(gk), c0, _U32), np.full(len(gk), c1, _U32))[0]
        L = M + (M & 1)
        h = L // 2
        c0 = np.arange(h, dtype=_U32)
        c1 = np.where(c0 + h < M, c0 + h, 0).astype(_U32)
        o0, o1 = _tf_multi(rk[:, None, :], c0[None, :], c1[None, :])
        bits = np.concatenate([o0, o1], axis=1)[:, :M]
    u = ((bits >> _U32(9)) | _U32(0x3F800000)).view(np.float32) - np.float32(1.0)
    sites = u < r[None, :]
    mask = np.logical_xor.accumulate(sites, axis=1).reshape(n, 2, M)
    gam = np.take_along_axis(parents, mask[..., None].astype(np.int64), axis=-1)[..., 0]
    return np.ascontiguousarray(gam.transpose(0, 2, 1)), keys[0].copy()


def _tf_multi(keys, c0, c1):
    """threefry with an array of keys (..., 2) broadcast against counters."""
    k0, k1 = keys[..., 0], keys[..., 1]
    ks = (k0, k1, k0 ^ k1 ^ _U32(0x1BD11BDA))
    x0 = c0 + ks[0]
    x1 = c1 + ks[1]
    for blk in range(5):
        rot = _R[4:] if blk & 1 else _R[:4]
        for r in rot:
            x0 = x0 + x1
            x1 = _rotl(x1, r)
            x1 = x1 ^ x0
        x0 = x0 + ks[(blk + 1) % 3]
        x1 = x1 + ks[(blk + 2) % 3] + _U32(blk + 1)
    return x0, x1


# ---- selection indices (reference breedgym/utils/index_functions.py) ---------------------------
def block_starts(B, chr_lens):
    """Marker offsets of the B blocks of every chromosome, ceil(len / B) markers each, the last ones
    cut at the chromosome end (index_functions.py:47-61).  Trailing blocks may be empty."""
    out, start = [0], 0
    for L in chr_lens:
        step, end = -(-int(L) // B), start + int(L)
        for _ in range(B):
            start = min(start + step, end)
            out.append(start)
    return np.asarray(out, np.int32)


def optimal_haploid_pop(pop, effects, block_start=None):
    """(n, M, 2) -> (n, M) bool (index_functions.py:36-78)."""
    pop = np.asarray(pop).astype(bool)
    eff = np.asarray(effects, np.float64).reshape(len(effects), -1)
    if block_start is None:
        pos = np.asarray(effects, np.float32).reshape(len(effects), -1)[:, 0] > 0
        return np.where(pos[None, :], pop.any(-1), pop.all(-1))
    ohp = np.empty(pop.shape[:2], bool)
    for s, e in zip(block_start[:-1], block_start[1:]):
        val = np.einsum("nmd,m->nd", pop[:, s:e].astype(np.float64), eff[s:e].sum(1))
        best = np.argmax(val, axis=-1)
        ohp[:, s:e] = np.take_along_axis(pop[:, s:e], best[:, None, None], axis=2)[..., 0]
    return ohp


def optimal_haploid_value(pop, effects, F=0.0, block_start=None):
    """index_functions.py:21-33: 2 * GEBV of the optimal haploid; the int(n * F) lowest-yield
    individuals (stable argsort of the first trait's GEBV) get -inf."""
    eff = np.asarray(effects, np.float64).reshape(len(effects), -1)
    ohv = 2.0 * (optimal_haploid_pop(pop, effects, block_start).astype(np.float64) @ eff)
    ohv = ohv.astype(np.float32).squeeze()
    if F > 0:
        ohv = np.array(ohv)
        ohv[poor_yield_indices(pop, effects, F)] = -np.inf
    return ohv


def poor_yield_indices(pop, effects, F):
    g = gebv(np.asarray(pop).astype(np.uint8), np.asarray(effects, np.float32).reshape(len(effects), -1))[:, 0]
    return np.argsort(g, kind="stable")[: int(len(pop) * F)]


def optimal_population_value(pop, effects, n, F=0.0, block_start=None):
    """index_functions.py:81-109: greedy choice of n individuals whose optimal haploids stack best.
    Returns the bool mask of the chosen individuals (the reference's "not OPV but a mask")."""
    pop = np.asarray(pop).astype(bool)
    eff = np.asarray(effects, np.float64).reshape(len(effects), -1)
    T = eff.shape[1]
    indices = np.delete(np.arange(len(pop)), poor_yield_indices(pop, effects, F))
    output = np.zeros(len(pop), bool)
    current = ~(np.asarray(effects, np.float32).reshape(len(effects), -1)[:, 0] > 0)
    G = optimal_haploid_pop(pop[indices], effects, block_start)
    for _ in range(n):
        dummy = np.stack((G, np.broadcast_to(current[None, :], G.shape)), axis=2)
        G = optimal_haploid_pop(dummy, effects, block_start)
        best = int(np.argmax((G.astype(np.float64) @ eff).astype(np.float32))) // T
        output[indices[best]] = True
        current = G[best]
        indices = np.delete(indices, best)
        G = optimal_haploid_pop(pop[indices], effects, block_start)
    assert np.count_nonzero(output) == n
    return output


def winner_layout(winners, n_local, world, rank, cap, pairs=None):
    """Population sharding (no counterpart in the reference, whose multi-device mode pipes whole observations,
    breedgym/vector/vec_env.py:150-236; include/breedhip.h, bh_winner_layout): from the replicated list of winners
    (global row indices, best first), rank r owning rows [r n_local, (r + 1) n_local) ->
    (send_idx (cap,), pos (k,), counts (world + 1,), idx_out or None), in plain array operations."""
    w = np.asarray(winners, np.int64).reshape(-1)
    owner = np.clip(w // n_local, 0, world - 1)
    hot = owner[:, None] == np.arange(world)[None, :]
    within = (np.cumsum(hot, axis=0) - hot)[np.arange(len(w)), owner] if len(w) else np.zeros(0, np.int64)
    counts = hot.sum(0)
    send = np.zeros(cap, np.int32)
    mine = (owner == rank) & (within < cap)
    send[within[mine]] = (w[mine] - rank * n_local).astype(np.int32)
    pos = (owner * cap + np.minimum(within, cap - 1)).astype(np.int32)
    out = None
    if pairs is not None:
        j = np.asarray(pairs, np.int64)
        j = np.clip(np.where(j < 0, j + len(w), j), 0, len(w) - 1)  # JAX's retrieval rule on the k winners
        out = pos[j].astype(np.int32)
    return send, pos, np.append(counts, counts.max(initial=0)).astype(np.int32), out
