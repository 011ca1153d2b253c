"""Randomised shapes through the cross entry points against the CPU oracle, for as long as the clock allows.

Each case: random (envs, pool rows, offspring per env, markers, traits), a random key mode (shared / independent keys, with
or without doubled haploids), one of JAX's two threefry layouts, a recombination vector with zeros, ones and chromosome
starts, sometimes a mutation probability; THREE generations on one context (the second and third find their masks made
ahead by the launch before -- or not, when the batch size changes between them), through bh_cross (byte layout) and
bh_cross_packed (bit planes + byte observation) side by side.  Checked per generation: offspring bytes and bit planes
bit-exact, the key chain, GEBV within 1e-5, the per-env reward.   usage: fuzz_cross.py [seconds] [seed]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from breedgym_amd import HipEngine
from oracle import c_oracle as co

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
co.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))


def pack_numpy(g):
    rows, M, _ = g.shape
    nw = (M + 63) // 64
    nw = nw if nw <= 256 else (nw + 7) // 8 * 8  # bh_packed_pitch: whole 128-byte lines for rows beyond 4 KB
    by = np.zeros((rows, nw * 64 * 2), np.uint8)
    by[:, : 2 * M] = g.reshape(rows, 2 * M)
    return np.packbits(by, axis=1, bitorder="little").view(np.uint16)


def one_case(case):
    M = int(rng.choice([rng.integers(1, 130), rng.integers(130, 3000), rng.integers(3000, 17000), rng.integers(17000, 40000)],
                       p=[0.3, 0.35, 0.25, 0.1]))
    T = int(rng.choice([0, 1, 2, 3, 5, 8, 9]))
    E = int(rng.integers(1, 6))
    n_pool = int(rng.integers(1, 40))
    layout = ("original", "partitionable")[int(rng.integers(0, 2))]
    mode = int(rng.integers(0, 4))  # bit 0: independent keys, bit 1: doubled haploids
    mutation = float(rng.choice([0.0, 0.0, 0.0, 0.02]))
    big = rng.random() < 0.15
    r = (rng.random(M) * 0.05).astype(np.float32)
    r[rng.random(M) < 0.2] = 0.0
    r[rng.random(M) < 0.02] = 1.0
    r[:: max(1, M // int(rng.integers(1, 8)))] = 0.5
    eff = rng.laplace(0, 6, (M, T)).astype(np.float32)
    eng = HipEngine(r, eff, threefry_layout=layout)
    co.set_threefry_layout(layout)
    if mutation:
        eng.set_mutation(mutation)
    pool = rng.integers(0, 2, (E, n_pool, M, 2), dtype=np.uint8)
    seed = int(rng.integers(0, 2**32))
    key_b, key_p, key_o = (co.key_from_seed(seed).copy() for _ in range(3))
    cur_b = eng.upload_population(pool.reshape(E * n_pool, M, 2))
    cur_p = eng.pack(cur_b)
    ref, n_in = pool, n_pool
    for gen in range(3):
        n_out = int(rng.integers(1, 2200 if big else 300))
        if M * E * n_out > 60_000_000:
            n_out = max(1, 60_000_000 // (M * E))
        idx = rng.integers(-n_in, n_in + 2, (E, n_out, 2), dtype=np.int32)
        if mode & 2:
            idx[..., 1] = idx[..., 0]  # doubled haploids: the same parent twice
        idx_d = torch.from_numpy(idx).to(eng.device)
        want_g = T > 0 and rng.random() < 0.85
        want_r = want_g and rng.random() < 0.5
        out_b = eng.empty_population(E * n_out)
        out_p, obs = eng.empty_packed(E * n_out), (eng.empty_population(E * n_out) if rng.random() < 0.7 else None)
        g_b = eng.empty((E, n_out, T), torch.float32) if want_g else None
        g_p = eng.empty((E, n_out, T), torch.float32) if want_g else None
        rw_b = eng.empty_reward(E) if want_r else None
        rw_p = eng.empty_reward(E) if want_r else None
        eng.cross(cur_b, E, n_in, idx_d, n_out, key_b, mode, out_b, g_b, rw_b)
        eng.cross_packed(cur_p, E, n_in, idx_d, n_out, key_p, mode, out_p, obs, g_p, rw_p)
        ref, key_o = co.cross(ref, idx, r, key_o, mode, mutation=mutation)
        what = f"case {case} gen {gen}: E={E} pool={n_in} n_out={n_out} M={M} T={T} mode={mode} {layout} mut={mutation} gebv={want_g} rew={want_r} obs={obs is not None}"
        got_b = eng.download_population(out_b).view(np.uint8).reshape(E, n_out, M, 2)
        assert np.array_equal(got_b, ref), what + ": byte-layout offspring"
        assert np.array_equal(out_p.cpu().numpy().view(np.uint16), pack_numpy(ref.reshape(E * n_out, M, 2))), what + ": bit planes"
        if obs is not None:
            assert np.array_equal(eng.download_population(obs).view(np.uint8).reshape(E, n_out, M, 2), ref), what + ": observation"
        assert np.array_equal(key_b, key_o) and np.array_equal(key_p, key_o), what + ": key chain"
        if want_g:
            g64 = co.gebv(ref, eff, want64=True)[1]
            tol = 1e-5 * np.maximum(np.abs(g64), 1.0)
            for name, g in (("bytes", g_b), ("bits", g_p)):
                assert np.all(np.abs(g.cpu().numpy().astype(np.float64) - g64) <= tol), what + f": GEBV ({name})"
            if want_r:
                best = g64.max(axis=(1, 2))
                for name, rw in (("bytes", rw_b), ("bits", rw_p)):
                    assert np.allclose(rw.cpu().numpy().astype(np.float64), best, rtol=1e-5, atol=1e-5), what + f": reward ({name})"
        cur_b, cur_p, n_in = out_b, out_p, n_out
    eng.sync()
    return what


t0, case, last = time.time(), 0, ""
while time.time() - t0 < budget:
    last = one_case(case)
    case += 1
    if case % 25 == 0:
        print(f"{case} cases, {time.time() - t0:.0f} s; last: {last}", flush=True)
co.set_threefry_layout("original")
print(f"fuzz_cross: {case} cases x 3 generations in {time.time() - t0:.0f} s, 0 mismatches")
