"""Selection indices with the names and meaning of the reference's
``breedgym/utils/index_functions.py``, computed on the device.

* ``yield_index``               first-trait GEBV (reference lines 6-10)
* ``optimal_haploid_pop``       best haploid each individual can pass on, marker-wise or in B blocks
                                per chromosome (lines 36-78)
* ``optimal_haploid_value``     OHV = 2 * GEBV of that haploid; the fraction F of lowest-yield
                                individuals is masked with -inf (lines 21-33)
* ``optimal_population_value``  greedy pick of n individuals whose haploids stack best; returns the
                                mask of the chosen, as the reference does (lines 81-109)

The reference runs these as NumPy loops on host copies; here the per-individual work is one kernel
launch (``bh_optimal_haploid``) and the greedy loop keeps its state on the GPU.  Each factory takes
the simulator's ``GEBV_model`` and returns a function of a population, like the reference's.
"""
import math
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from ..engine import DeviceArray


def yield_index(GEBV_model) -> Callable:
    return lambda population: GEBV_model(population)[..., 0]


def block_offsets(B: int, chr_lens: Sequence[int]) -> np.ndarray:
    """Marker offsets of B blocks per chromosome, ceil(len / B) markers each, cut at the chromosome
    end (reference lines 47-61).  The last blocks of a short chromosome may be empty."""
    offsets, at = [0], 0
    for length in chr_lens:
        step, stop = math.ceil(int(length) / B), at + int(length)
        for _ in range(B):
            at = min(at + step, stop)
            offsets.append(at)
    return np.asarray(offsets, np.int32)


def _device_blocks(sim, B, chr_lens) -> Optional[torch.Tensor]:
    if B is None:
        return None
    if chr_lens is None:
        raise AssertionError("chr_lens needed with B")  # the reference asserts the same
    return torch.from_numpy(block_offsets(B, chr_lens)).to(sim.engine.device)


def _haploids(GEBV_model, population, blocks, want_pop=True, want_value=True):
    sim = GEBV_model._sim
    rows, lead = sim._rows_of(population)
    if len(lead) != 1:
        raise ValueError("index functions take one population (n, M, 2)")
    return sim.engine.optimal_haploid(rows, blocks, want_pop, want_value)


def optimal_haploid_pop(GEBV_model, population, B: Optional[int] = None, chr_lens=None) -> DeviceArray:
    """(n, M, 2) genotypes -> (n, M) bool optimal haploids."""
    blocks = _device_blocks(GEBV_model._sim, B, chr_lens)
    ohp, _ = _haploids(GEBV_model, population, blocks, want_value=False)
    return DeviceArray.wrap(ohp.view(torch.bool))


def _poor_yield_indices(GEBV_model, population, F: float) -> torch.Tensor:
    """The int(n * F) individuals with the lowest first-trait GEBV (stable order on ties)."""
    n_poor = int(len(population) * F)
    yields = yield_index(GEBV_model)(population).as_subclass(torch.Tensor)
    return torch.argsort(yields, stable=True)[:n_poor]


def optimal_haploid_value(GEBV_model, F: float = 0, B: Optional[int] = None, chr_lens=None) -> Callable:
    def optimal_haploid_value_f(population) -> np.ndarray:
        blocks = _device_blocks(GEBV_model._sim, B, chr_lens)
        _, ohv = _haploids(GEBV_model, population, blocks, want_pop=False)
        if F > 0:
            ohv[_poor_yield_indices(GEBV_model, population, F)] = -float("inf")
        return ohv.cpu().numpy().squeeze()  # the reference returns a host array too (line 26)

    return optimal_haploid_value_f


def optimal_population_value(GEBV_model, n: int, F: float = 0, B: Optional[int] = None, chr_lens=None) -> Callable:
    def optimal_population_value_f(population) -> np.ndarray:
        sim = GEBV_model._sim
        eng = sim.engine
        blocks = _device_blocks(sim, B, chr_lens)
        rows, lead = sim._rows_of(population)
        if len(lead) != 1:
            raise ValueError("index functions take one population (n, M, 2)")
        total, M, T = lead[0], sim.n_markers, GEBV_model.n_traits
        keep = torch.ones(total, dtype=torch.bool, device=eng.device)
        keep[_poor_yield_indices(GEBV_model, population, F)] = False
        candidates = torch.nonzero(keep).reshape(-1)  # ascending, like np.delete on arange
        chosen = torch.zeros(total, dtype=torch.bool, device=eng.device)
        # the haploid the chosen set can already assemble: starts as the worst one (line 91)
        stack = torch.from_numpy(~(GEBV_model.marker_effects[:, 0] > 0)).to(eng.device).view(torch.uint8)
        own, _ = eng.optimal_haploid(rows, blocks, want_value=False)  # (total, M): never changes per individual
        for _ in range(n):
            # candidate haploid next to the stack, as one diploid "individual" each (lines 95-96)
            G = own[candidates]
            paired = torch.stack((G, stack.expand_as(G)), dim=2).reshape(len(candidates), M, 2)
            merged, value = eng.optimal_haploid(eng.pack_device(paired), blocks)
            # the haploid is scored with ploidy 1 (line 98): half of the diploid-style value
            best = int(torch.argmax(value.reshape(-1))) // T
            chosen[candidates[best]] = True
            stack = merged[best]
            candidates = torch.cat((candidates[:best], candidates[best + 1:]))
        if int(chosen.sum()) != n:
            raise AssertionError("optimal_population_value did not pick n individuals")
        return chosen.cpu().numpy()  # "not OPV but a mask" (line 107)

    return optimal_population_value_f
