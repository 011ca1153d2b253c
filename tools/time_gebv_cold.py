"""The one-trait GEBV pass over 8192 bit-plane rows (configs[2]'s step: gebv_bits_kernel<1, 4>) with its rows in the
Infinity Cache (the same 205 MB read again and again), out of it (twelve buffers in turn: 2.5 GB), and behind 2 GB of
stores (what the step's long-row launch leaves) -- for rows whose pitch is and is not a whole number of 128-byte lines.
usage: python tools/time_gebv_cold.py [reps]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from bench import time_events  # noqa: E402
from breedgym_amd import HipEngine  # noqa: E402
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
R = 8192
for M in (100_000, 100_352, 99_840):
    gm = synthetic_genetic_map(M, n_chr=21, n_traits=1, seed=1)
    eng = HipEngine(gm.recombination_vec, gm.marker_effects)
    small = eng.pack(eng.upload_population(synthetic_genotypes(512, M, seed=0)))
    one = torch.cat([small] * (R // 512)).contiguous()
    g = eng.empty((R, 1), torch.float32)
    pitch = one.shape[1] * one.element_size()
    bufs = [one.clone() for _ in range(12)]
    hot = time_events(lambda i: eng.gather_packed(one, None, R, None, None, g), reps)
    rot = time_events(lambda i: eng.gather_packed(bufs[i % 12], None, R, None, None, g), reps)
    flush = torch.empty(2 << 30, dtype=torch.uint8, device=one.device)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps):
        flush.fill_(i & 255)
        ev[i][0].record()
        eng.gather_packed(one, None, R, None, None, g)
        ev[i][1].record()
    torch.cuda.synchronize()
    behind = float(np.median([a.elapsed_time(b) for a, b in ev]))
    mb = R * pitch / 1e6
    print(f"M {M} pitch {pitch} B (mod 128: {pitch % 128}): rows {mb:.1f} MB | same rows again {hot * 1e3:.1f} us "
          f"({mb / hot / 1e3:.2f} TB/s) | twelve buffers in turn {rot * 1e3:.1f} us ({mb / rot / 1e3:.2f} TB/s) | "
          f"behind a 2 GB fill (events around the one launch, median) {behind * 1e3:.1f} us", flush=True)
    del bufs, flush, one, eng
