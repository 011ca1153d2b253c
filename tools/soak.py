"""Soak: many generations with the byte-layout and the bit-plane kernels side by side (same seeds),
compared every generation on the device, and against the CPU oracle every `check` generations.
Shapes are chosen to go through the run-ahead masks, the mask turns (four waves per cross, more keyed
crosses than row workgroups), the long-row launch (LDS-DMA stream, rows in turns) and both key modes."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import HipEngine
from oracle import c_oracle as co

def soak(M, E, n, mode, gens, check, seed):
    rng = np.random.default_rng(seed)
    r = (rng.random(M) * 0.004).astype(np.float32); r[:: max(1, M // 9)] = 0.5
    eff = rng.laplace(0, 6, (M, 1)).astype(np.float32)
    eng_b, eng_p = HipEngine(r, eff), HipEngine(r, eff)
    pool = rng.integers(0, 2, (E * n, M, 2), dtype=np.uint8)
    rows_b = eng_b.upload_population(pool); rows_p = eng_p.pack(eng_p.upload_population(pool))
    kb, kp, ko = co.key_from_seed(seed).copy(), co.key_from_seed(seed).copy(), co.key_from_seed(seed)
    ref = pool.reshape(E, n, M, 2)
    bad = 0
    for g in range(gens):
        idx = rng.integers(0, n, (E, n, 2), dtype=np.int32)
        d_idx = torch.from_numpy(idx).cuda()
        out_b, gb = eng_b.empty_population(E * n), eng_b.empty((E, n, 1), torch.float32)
        eng_b.cross(rows_b, E, n, d_idx, n, kb, mode, out_b, gb)
        out_p, obs, gp = eng_p.empty_packed(E * n), eng_p.empty_population(E * n), eng_p.empty((E, n, 1), torch.float32)
        eng_p.cross_packed(rows_p, E, n, d_idx, n, kp, mode, out_p, obs, gp)
        if not torch.equal(obs, out_b) or not np.array_equal(kb, kp) or not torch.allclose(gb, gp, rtol=1e-6, atol=1e-4):
            bad += 1; print(f"  MISMATCH bytes vs bits at generation {g}")
        if check and g % check == 0:
            ref, ko = co.cross(ref, idx, r, ko, mode)
            got = eng_b.download_population(out_b).view(np.uint8).reshape(E, n, M, 2)
            if not np.array_equal(got, ref) or not np.array_equal(ko, kb):
                bad += 1; print(f"  MISMATCH vs oracle at generation {g}")
        elif check:
            ref, ko = None, None  # oracle chain not followed every generation: re-anchor below
        rows_b, rows_p = out_b, out_p
        if check and ref is None:
            ref = eng_b.download_population(out_b).view(np.uint8).reshape(E, n, M, 2); ko = kb.copy()
    print(f"M={M} E={E} n={n} mode={mode}: {gens} generations, {bad} mismatches")
    return bad

bad = 0
bad += soak(10_000, 16, 256, 0, 300, 25, 1)
bad += soak(10_000, 16, 256, 1, 100, 25, 2)
bad += soak(16_400, 2, 1100, 0, 60, 10, 3)     # MW = 4, 1100 keyed crosses vs 275 row workgroups: turns
bad += soak(16_400, 1, 1300, 1, 60, 10, 4)
bad += soak(50_000, 4, 600, 0, 40, 10, 5)
bad += soak(1_000, 8, 4100, 1, 60, 10, 6)      # one wave per cross, 32,800 keyed crosses
# many long rows: the LDS-DMA stream next to the stream-per-lane mask role (rows in turns, 5 / 5 / 9 / 25 chunks per row)
bad += soak(16_523, 2, 4200, 0, 30, 10, 7)
bad += soak(20_000, 1, 8200, 1, 20, 10, 8)
bad += soak(33_000, 4, 2100, 0, 20, 10, 9)
bad += soak(100_000, 4, 2048, 0, 12, 6, 10)
sys.exit(1 if bad else 0)
