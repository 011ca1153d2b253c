"""GEBV from bit-plane rows (gebv_bits.h) at BASELINE configs[4] and at the configs[2] row count: HIP-event time per call."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import HipEngine
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

def timed(f, reps=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

M = 100_000
base = synthetic_genotypes(2048, M, seed=0)
for T in (1, 2, 3, 5, 8, 10):
    gm = synthetic_genetic_map(M, n_chr=21, n_traits=T, seed=1)
    eng = HipEngine(gm.recombination_vec, gm.marker_effects)
    rows = eng.upload_population(base)
    packed = eng.pack(rows)
    for n in (2048, 8192):
        src = packed if n == 2048 else packed.repeat(4, 1).contiguous()
        g = eng.empty((n, T), torch.float32)
        us = timed(lambda: eng.gather_packed(src, None, n, None, None, g))
        print(f"T={T:2d} rows={n}: bit planes {us:8.1f} us  ({n * eng.packed_pitch / us / 1e3:7.0f} GB/s of bit-plane bytes)", flush=True)
    us = timed(lambda: eng.gebv(rows))
    print(f"T={T:2d} rows=2048: byte rows  {us:8.1f} us  ({2048 * 2 * M / us / 1e3:7.0f} GB/s of genotype bytes)", flush=True)
    del eng
