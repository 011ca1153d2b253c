"""bh_optimal_haploid (the OHV / OPV selection indices' kernel) at BASELINE sizes: per call, by HIP events.
usage: python tools/time_ohv.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from bench import time_events  # noqa: E402
from breedgym_amd import HipEngine  # noqa: E402
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes  # noqa: E402
from oracle import numpy_oracle as no  # noqa: E402  (block offsets only: a tool, not the product)

for N, M, T in ((370, 10_000, 1), (2048, 100_000, 1), (2048, 100_000, 8)):
    gm = synthetic_genetic_map(M, n_chr=21 if M > 10_000 else 10, n_traits=T, seed=1)
    eng = HipEngine(gm.recombination_vec, gm.marker_effects)
    rows = eng.upload_population(synthetic_genotypes(N, M, seed=0))
    bs = torch.from_numpy(no.block_starts(5, gm.chr_lens)).to(eng.device)
    for name, b in (("no blocks", None), ("5 blocks per chromosome", bs)):
        for want_pop in (True, False):
            ms = time_events(lambda i: eng.optimal_haploid(rows, b, want_pop=want_pop), 20, warm=2)
            print(f"{N} x {M} x {T} traits, {name}, haploid {'written' if want_pop else 'not written'}: {ms * 1e3:9.1f} us "
                  f"({N * 2 * M / ms / 1e6:7.1f} GB/s of genotype bytes)", flush=True)
