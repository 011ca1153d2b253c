"""Randomised founder draws, top-k calls and SelectionScores conversions against the CPU oracle, for as long as the clock
allows: sizes on both sides of every route boundary of csrc/select.hip (one workgroup in LDS up to 8192 elements, the
wanted-only shuffle, the radix sort in HBM; top-k by one workgroup's sort, radix select, chunk levels, the sort), both
threefry layouts, scores that are mostly ties.  Everything bit-exact.   usage: fuzz_select.py [seconds] [seed]"""
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
engines = {lay: HipEngine(np.full(64, 0.01, np.float32), np.ones((64, 1), np.float32), threefry_layout=lay)
           for lay in ("original", "partitionable")}


def scores(shape):
    kind = rng.integers(0, 3)
    if kind == 0:
        s = rng.normal(0, 1, shape)
    elif kind == 1:
        s = np.round(rng.normal(0, 2, shape))  # a handful of distinct values: ties everywhere
    else:
        s = rng.normal(0, 1, shape)
        s[rng.random(shape) < 0.3] = 0.0
        s[rng.random(shape) < 0.1] = -0.0
    s = s.astype(np.float32)
    flat = s.reshape(-1)
    if flat.size > 4 and rng.random() < 0.3:
        flat[rng.integers(0, flat.size, 3)] = np.inf
        flat[rng.integers(0, flat.size, 3)] = -np.inf
    return s


def pick(*ranges):
    lo, hi = ranges[int(rng.integers(0, len(ranges)))]
    return int(rng.integers(lo, hi + 1))


def one_case(case):
    lay = ("original", "partitionable")[int(rng.integers(0, 2))]
    eng = engines[lay]
    co.set_threefry_layout(lay)
    kind = int(rng.integers(0, 3))
    seed = int(rng.integers(0, 2**32))
    if kind == 0:  # founders: choice(N, n, replace=False) per env
        N = pick((2, 600), (600, 8192), (8193, 9000), (9000, 270_000), (262_140, 262_150), (270_000, 700_000))
        n = pick((1, min(N, 50)), (1, N), (max(1, N // 17), max(1, N // 15)), (N, N))
        E = pick((1, 1), (1, 5)) if N * 1 < 200_000 else 1
        if E * n > 3_000_000:
            n = 3_000_000 // E
        off, total = (int(rng.integers(0, 4)), 0)
        total = off + E + int(rng.integers(0, 3)) if rng.random() < 0.3 else 0
        eng.set_env_window(off if total else 0, total)
        got, key = eng.founders(co.key_from_seed(seed), E, N, n)
        eng.set_env_window(0, 0)
        if total:  # the shard's envs are envs [off, off + E) of `total`: the oracle draws the whole and is cut
            ref_all, ref_key = co.founders(co.key_from_seed(seed), total, N, n)
            ref = ref_all[off:off + E]
        else:
            ref, ref_key = co.founders(co.key_from_seed(seed), E, N, n)
        what = f"case {case}: founders E={E} N={N} n={n} window=({off},{total}) {lay}"
        assert np.array_equal(got.cpu().numpy().reshape(E, n), ref), what
        assert np.array_equal(key, ref_key), what + ": key"
    elif kind == 1:  # top-k
        rows = pick((1, 1), (1, 6))
        n = pick((1, 300), (300, 4096), (4097, 16384), (16385, 70_000), (70_000, 400_000))
        if rows * n > 1_500_000:
            rows = 1
        k = pick((1, min(n, 30)), (1, n), (max(1, n // 10), max(1, n // 10)), (max(1, n // 2), min(n, n // 2 + 2)), (n, n))
        s = scores((rows, n))
        got = eng.topk(torch.from_numpy(s).to(eng.device), k).cpu().numpy()
        what = f"case {case}: topk rows={rows} n={n} k={k}"
        for r in range(rows):
            assert np.array_equal(got[r], co.topk(s[r], k)), what + f" row {r}"
    else:  # SelectionScores._convert_actions
        n = pick((3, 200), (200, 3000), (3000, 9000))
        k = pick((2, min(n, 12)), (2, min(n, 130)), (max(2, n // 10), max(2, n // 10)))
        nc = pick((1, min(k * (k - 1) // 2, n)), (min(k * (k - 1) // 2, n), min(k * (k - 1) // 2, n)))
        E = pick((1, 1), (1, 5))
        s = scores((E, n))
        got, key = eng.selection_actions(torch.from_numpy(s).to(eng.device), k, nc, co.key_from_seed(seed))
        ref, ref_key = co.selection_actions(s, k, nc, co.key_from_seed(seed))
        what = f"case {case}: selection_actions E={E} n={n} k={k} n_crosses={nc} {lay}"
        assert np.array_equal(got.cpu().numpy(), ref), what
        assert np.array_equal(key, ref_key), what + ": key"
    return what


t0, case, last = time.time(), 0, ""
while time.time() - t0 < budget:
    last = one_case(case)
    case += 1
    if case % 50 == 0:
        print(f"{case} cases, {time.time() - t0:.0f} s; last: {last}", flush=True)
for e in engines.values():
    e.sync()
co.set_threefry_layout("original")
print(f"fuzz_select: {case} cases in {time.time() - t0:.0f} s, 0 mismatches")
