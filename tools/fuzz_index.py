"""Randomised bh_optimal_haploid (with and without blocks, blocks that end inside a 16-byte vector, empty blocks, equal
haplotypes, zero effects), unpack, and gathers of bit-plane rows with the GEBV on the matrix cores (many rows out of few
included) against the CPU oracle, for as long as the clock allows.   usage: fuzz_index.py [seconds] [seed]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from breedgym_amd import HipEngine
from oracle import c_oracle as co
from oracle import numpy_oracle as no

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def pick(*ranges):
    lo, hi = ranges[int(rng.integers(0, len(ranges)))]
    return int(rng.integers(lo, hi + 1))


def close(got, ref64):
    return np.all(np.abs(got.astype(np.float64) - ref64) <= 1e-5 * np.maximum(np.abs(ref64), 1.0))


def one_case(case):
    M = pick((1, 130), (130, 3000), (3000, 17000), (16380, 16450), (17000, 40000))
    N = pick((1, 40), (40, 200)) if M < 17000 else pick((1, 12), (30, 70))
    T = int(rng.choice([1, 2, 3, 5, 8, 9]))
    pop = rng.integers(0, 2, (N, M, 2), dtype=np.uint8)
    if N > 2 and rng.random() < 0.5:
        pop[1] = pop[0]
        pop[2, :, 1] = pop[2, :, 0]  # every block a tie: the first haplotype
    eff = rng.laplace(0, 6, (M, T)).astype(np.float32)
    eff[rng.random(M) < 0.1] = 0.0
    eng = HipEngine(np.full(M, 0.01, np.float32), eff)
    rows = eng.upload_population(pop)
    what = f"case {case}: N={N} M={M} T={T}"
    kind = int(rng.integers(0, 3))
    if kind == 0:  # the selection indices
        n_chr = pick((1, 1), (2, 7))
        cuts = np.sort(rng.choice(np.arange(1, M), size=min(n_chr - 1, M - 1), replace=False)) if M > 1 else np.array([], int)
        lens = np.diff(np.concatenate(([0], cuts, [M]))).tolist()
        B = None if rng.random() < 0.35 else pick((1, 6), (6, 60), (60, 900), (900, 1700))
        bs = None if B is None else no.block_starts(B, lens)
        dbs = None if bs is None else torch.from_numpy(bs).to(eng.device)
        what += f" OHV B={B} chr={len(lens)} blocks={0 if bs is None else len(bs) - 1}"
        want_pop, want_value = bool(rng.random() < 0.8), bool(rng.random() < 0.8)
        if not (want_pop or want_value):
            want_pop = True
        ohp, ohv = eng.optimal_haploid(rows, dbs, want_pop=want_pop, want_value=want_value)
        ref_p, ref_v = co.optimal_haploid(pop, eff, bs)
        ok = (not want_pop or np.array_equal(ohp.cpu().numpy(), ref_p)) and \
             (not want_value or np.allclose(ohv.cpu().numpy(), ref_v, rtol=1e-6, atol=1e-4))
    elif kind == 1:  # unpack, through an index or not
        packed = eng.pack(rows)
        R = pick((1, 3 * N), (N, N))
        idx = None if rng.random() < 0.4 else torch.from_numpy(rng.integers(-N - 1, N + 2, R, dtype=np.int32)).to(eng.device)
        R = N if idx is None else R
        out = eng.empty_population(R)
        eng.gather_packed(packed, idx, R, None, out)
        ii = np.arange(N) if idx is None else idx.cpu().numpy().astype(np.int64)
        ii = np.clip(np.where(ii < 0, ii + N, ii), 0, N - 1)
        what += f" unpack R={R} idx={idx is not None}"
        ok = np.array_equal(eng.download_population(out).view(np.uint8), pop[ii])
    else:  # gathers of bit-plane rows with GEBV
        packed = eng.pack(rows)
        R = pick((1, N), (2 * N, 6 * N), (32, 200))
        idx = torch.from_numpy(rng.integers(-N - 1, N + 2, R, dtype=np.int32)).to(eng.device)
        ii = idx.cpu().numpy().astype(np.int64)
        ii = np.clip(np.where(ii < 0, ii + N, ii), 0, N - 1)
        want_p, want_b = bool(rng.random() < 0.6), bool(rng.random() < 0.6)
        outp = eng.empty_packed(R) if want_p else None
        outb = eng.empty_population(R) if want_b else None
        g = eng.empty((R, T), torch.float32)
        eng.gather_packed(packed, idx, R, outp, outb, g)
        what += f" gather R={R} packed={want_p} bytes={want_b}"
        ok = close(g.cpu().numpy(), co.gebv(pop[ii], eff, want64=True)[1])
        if want_b:
            ok = ok and np.array_equal(eng.download_population(outb).view(np.uint8), pop[ii])
        if want_p:
            back = eng.empty_population(R)
            eng.gather_packed(outp, None, R, None, back)
            ok = ok and np.array_equal(eng.download_population(back).view(np.uint8), pop[ii])
    eng.close()
    return ok, what


t0, case, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    ok, what = one_case(case)
    if not ok:
        bad += 1
        print("MISMATCH", what, flush=True)
    case += 1
    if case % 200 == 0:
        print(f"{case} cases, {time.time() - t0:.0f} s; last: {what}", flush=True)
print(f"fuzz_index: {case} cases in {time.time() - t0:.0f} s, {bad} mismatches")
sys.exit(1 if bad else 0)
