"""Seeded synthetic inputs with the statistics of the reference's sample data.

The reference's 370 x 10k default population (``small_geno.npy``) is a stripped large blob
(``.MISSING_LARGE_BLOBS``), so benchmarks and size tests synthesise one:
  genotypes : per-marker allele frequency ~ Beta(0.5, 2) (mean 0.2; the shipped 370 x 1000
              sample has mean 0.176), individuals mostly inbred (second allele = first with
              probability 0.97; measured heterozygosity 0.027)
  map       : chromosomes with sizes in the proportions of ``small_genetic_map.txt``;
              RecombRate zero w.p. 0.16, else log-normal scaled to sum 1.5 per chromosome
              (measured: 14.97 over 10 chromosomes); first marker of a chromosome = 0.5
  effects   : Laplace with std 9 (heavy tailed like the measured Yield column)
"""
import numpy as np

from .genetic_map import GeneticMap


def synthetic_genotypes(n: int, n_markers: int, seed: int = 0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    p = rng.beta(0.5, 2.0, size=n_markers).astype(np.float32)
    a0 = rng.random((n, n_markers), dtype=np.float32) < p
    a1 = a0 ^ (rng.random((n, n_markers), dtype=np.float32) < 0.03)  # 3 % heterozygous sites
    return np.stack((a0, a1), axis=-1)


def synthetic_genetic_map(n_markers: int, n_chr: int = 10, n_traits: int = 1, seed: int = 1) -> GeneticMap:
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.7, 1.5, n_chr)
    lens = np.maximum(1, np.floor(w / w.sum() * n_markers).astype(np.int64))
    lens[-1] += n_markers - lens.sum()
    r = np.empty(n_markers, np.float64)
    start = 0
    for L in lens:
        x = rng.lognormal(0.0, 1.0, L)
        x[rng.random(L) < 0.16] = 0.0
        x *= 1.5 / max(x.sum(), 1e-12)
        r[start : start + L] = np.minimum(x, 0.5)
        r[start] = 0.5
        start += L
    eff = rng.laplace(0.0, 9.0 / np.sqrt(2.0), (n_markers, n_traits)).astype(np.float32)
    names = ["Yield"] if n_traits == 1 else [f"trait{t}" for t in range(n_traits)]
    return GeneticMap(r.astype(np.float32), eff, names, lens)
