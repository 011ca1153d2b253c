"""Genetic-map ingestion: TSV / DataFrame -> recombination vector, marker effects, chromosome runs.

On-disk schema (headers of ``breedgym/data/*.txt|csv`` in the reference): one row per marker,
a chromosome column (``CHR.PHYS``), either a per-marker recombination probability
(``RecombRate``) or a genetic position in centimorgan (``cM``), and one column per trait.
DataFrame input as in ``scripts/time_wheat.py:74-81``.

[EXTERNAL/UNVERIFIED] The derivation rules are chromax's (the reference passes the map straight
to ``chromax.Simulator``, vec_env.py:45): traits = every column that is not the chromosome /
position / recombination column; ``r`` = ``RecombRate`` if present, else the cM difference to
the previous marker / 100; the first marker of every chromosome gets r = 0.5 (independent
assortment of chromosomes).
"""
from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Sequence, Union

import numpy as np
import pandas as pd


@dataclass
class GeneticMap:
    recombination_vec: np.ndarray  # (M,) float32
    marker_effects: np.ndarray  # (M, T) float32
    trait_names: List[str]
    chr_lens: np.ndarray  # (n_chr,) int
    positions_cM: Optional[np.ndarray] = None

    @property
    def n_markers(self) -> int:
        return len(self.recombination_vec)


def read_genetic_map(
    genetic_map: Union[str, Path, pd.DataFrame],
    trait_names: Optional[Sequence[str]] = None,
    chr_column: str = "CHR.PHYS",
    position_column: str = "cM",
    recombination_column: str = "RecombRate",
) -> GeneticMap:
    if isinstance(genetic_map, pd.DataFrame):
        df = genetic_map
    else:
        df = pd.read_table(genetic_map, sep="\t")
        if df.shape[1] == 1:  # not tab separated after all
            df = pd.read_csv(genetic_map, sep=None, engine="python")
    if chr_column not in df.columns:
        raise ValueError(f"genetic map has no chromosome column {chr_column!r}")

    other = {chr_column, position_column, recombination_column}
    if trait_names is None:
        trait_names = [c for c in df.columns if c not in other]
    else:
        trait_names = list(trait_names)
        missing = [t for t in trait_names if t not in df.columns]
        if missing:
            raise ValueError(f"traits {missing} are not columns of the genetic map")

    chrom = df[chr_column].to_numpy()
    n = len(chrom)
    first = np.ones(n, dtype=bool)
    first[1:] = chrom[1:] != chrom[:-1]
    starts = np.flatnonzero(first)
    chr_lens = np.diff(np.append(starts, n))

    cM = None
    if recombination_column in df.columns:
        r = df[recombination_column].to_numpy(dtype=np.float64).copy()
    elif position_column in df.columns:
        cM = df[position_column].to_numpy(dtype=np.float64)
        r = np.zeros(n, np.float64)
        r[1:] = (cM[1:] - cM[:-1]) / 100.0
    else:
        raise ValueError(
            f"genetic map needs a {recombination_column!r} or a {position_column!r} column"
        )
    r[first] = 0.5
    effects = df[trait_names].to_numpy(dtype=np.float32) if trait_names else np.zeros((n, 0), np.float32)
    return GeneticMap(
        recombination_vec=r.astype(np.float32),
        marker_effects=np.ascontiguousarray(effects, np.float32),
        trait_names=list(trait_names),
        chr_lens=chr_lens,
        positions_cM=cM,
    )
