"""Every data-moving entry point of the C ABI once at two BASELINE shapes, with the bytes it must move and the time it takes
(HIP events, 20 calls): a scan for kernels nobody has looked at with a clock.   usage: python tools/scan_ops.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from bench import time_events  # noqa: E402
from breedgym_amd import HipEngine  # noqa: E402
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes  # noqa: E402


def line(name, ms, nbytes):
    print(f"  {name:62s} {ms * 1e3:9.1f} us  {nbytes / 1e6:9.1f} MB  {nbytes / ms / 1e9:7.2f} TB/s", flush=True)


for N, M, T in ((4096, 10_000, 1), (2048, 100_000, 1), (2048, 100_000, 3), (256, 10_000, 3)):
    print(f"{N} rows x {M} markers x {T} traits")
    gm = synthetic_genetic_map(M, n_chr=10, n_traits=T, seed=1)
    eng = HipEngine(gm.recombination_vec, gm.marker_effects)
    geno = synthetic_genotypes(min(N, 512), M, seed=0)
    rows = torch.cat([eng.upload_population(geno)] * (N // min(N, 512))).contiguous()
    packed = eng.pack(rows)
    pitch, pp = eng.pitch, eng.packed_pitch
    idx = torch.randint(0, N, (N,), dtype=torch.int32, device=eng.device)
    out_b, out_p = eng.empty_population(N), eng.empty_packed(N)
    g = eng.empty((N, T), torch.float32)
    line("pack (byte rows -> bit planes)", time_events(lambda i: eng.pack(rows, out_p), 20, warm=2), N * (pitch + pp))
    line("unpack (bit planes -> byte rows)", time_events(lambda i: eng.unpack(packed, out_b), 20, warm=2), N * (pitch + pp))
    line("gather, byte rows through an index", time_events(lambda i: eng.gather(rows, idx, out_b), 20, warm=2), N * 2 * pitch)
    line("gather, byte rows through an index + GEBV", time_events(lambda i: eng.gather(rows, idx, out_b, g), 20, warm=2), N * 2 * pitch)
    line("gather, bit planes through an index -> bit planes", time_events(lambda i: eng.gather_packed(packed, idx, N, out_p), 20, warm=2), N * 2 * pp)
    line("gather, bit planes -> bit planes + byte rows + GEBV", time_events(lambda i: eng.gather_packed(packed, idx, N, out_p, out_b, g), 20, warm=2), N * (2 * pp + pitch))
    line("GEBV of byte rows", time_events(lambda i: eng.gebv(rows, g), 20, warm=2), N * pitch)
    line("GEBV of bit planes", time_events(lambda i: eng.gather_packed(packed, None, N, None, None, g), 20, warm=2), N * pp)
    key = np.array([1, 2], np.uint32)
    for Ec in (1, 4):
        n_out = N // Ec
        ci = torch.randint(0, n_out, (Ec, n_out, 2), dtype=torch.int32, device=eng.device)
        for obs in (True, False):
            def cross(i, obs=obs):
                eng.cross_packed(packed, Ec, n_out, ci, n_out, key, 0, out_p, out_b if obs else None, g)
            line(f"cross, {Ec} env(s) x {n_out}, shared keys, GEBV, {'with' if obs else 'no'} byte rows out",
                 time_events(cross, 20, warm=3), N * (3 * pp + (pitch if obs else 0)))
    E = 16 if N % 16 == 0 else 1
    r = eng.empty_reward(E)
    line(f"reward max ({E} envs)", time_events(lambda i: eng.reward_max(g, E, r), 20, warm=2), N * T * 4)
    host = np.ascontiguousarray(geno)
    import time
    t0 = time.perf_counter()
    for _ in range(3):
        up = eng.upload_population(host)
    torch.cuda.synchronize()
    print(f"  upload of {host.shape[0]} host rows (validated): {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms for {host.nbytes / 1e6:.1f} MB")
    t0 = time.perf_counter()
    for _ in range(3):
        eng.download_population(up)
    print(f"  download of the same: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
    del eng, rows, packed, out_b, out_p
