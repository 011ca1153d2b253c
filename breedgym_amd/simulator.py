"""``Simulator``: the object the reference reaches through ``self.simulator``.

The reference's environments call exactly these members of ``chromax.Simulator`` on the hot
path: the constructor, ``.device``, ``.load_population``, ``.set_seed``,
``.cross``, ``.GEBV_model`` (vec_env.py:45-50,77,114,133), plus ``.GEBV``, ``.corrcoef``,
``.select``, ``._diallel_indices`` from the single env and the wrappers (breedgym.py:233,240;
wrappers.py:75,80; vec_wrappers.py:66).  This class offers the same names with the same
argument meaning, backed by the HIP engine.  ``cross_indexed`` is the fused form the
environments use: it takes the population and the parent *indices* and never materialises
``population[actions]``.
"""
from pathlib import Path
from typing import Callable, Optional, Sequence, Tuple, Union

import numpy as np
import pandas as pd
import torch

from . import _lib, rng
from .engine import DeviceArray, HipEngine
from .genetic_map import GeneticMap, read_genetic_map


class TraitModel:
    """``simulator.GEBV_model``: callable marker-effect model (additive, no offset).

    GEBV[..., t] = sum_m (pop[..., m, 0] + pop[..., m, 1]) * marker_effects[m, t]
    (scripts/optimize_diff_cross.py:16-17, utils/index_functions.py:55)."""

    def __init__(self, simulator: "Simulator", marker_effects: np.ndarray):
        self._sim = simulator
        self.marker_effects = marker_effects  # (M, T) float32, host
        self.positive_mask = marker_effects > 0  # utils/index_functions.py:70
        self.n_traits = marker_effects.shape[1]

    def __call__(self, population) -> DeviceArray:
        rows, lead = self._sim._rows_of(population)
        out = self._sim.engine.gebv(rows)
        return DeviceArray.wrap(out.reshape(*lead, self.n_traits))


class Simulator:
    def __init__(
        self,
        genetic_map: Union[str, Path, pd.DataFrame, GeneticMap],
        trait_names: Optional[Sequence[str]] = None,
        chr_column: str = "CHR.PHYS",
        position_column: str = "cM",
        recombination_column: str = "RecombRate",
        mutation: float = 0.0,
        h2=None,
        seed: Optional[int] = None,
        device=None,
        backend=None,
        *,
        threefry_layout: str = "original",
        engine=None,
    ):
        if not 0.0 <= float(mutation) <= 1.0:
            raise ValueError("mutation is a probability")
        gm = genetic_map if isinstance(genetic_map, GeneticMap) else read_genetic_map(
            genetic_map, trait_names, chr_column, position_column, recombination_column
        )
        self.genetic_map = gm
        self.trait_names = gm.trait_names
        self.n_markers = gm.n_markers
        self.chr_lens = gm.chr_lens
        self.recombination_vec = gm.recombination_vec
        self.mutation = mutation
        # `engine=` is the seam tests use to drive the host logic without a GPU; product code
        # never passes it, so the HIP engine (and its loud failure without a device) is the default.
        _lib.layout_code(threefry_layout)  # validates
        self.threefry_layout = threefry_layout
        if engine is not None:
            self.engine = engine
            if threefry_layout != "original":
                engine.set_threefry_layout(threefry_layout)
        else:
            self.engine = HipEngine(gm.recombination_vec, gm.marker_effects, device, threefry_layout)
        if mutation != 0.0:  # chromax's per-allele flip probability; 0 keeps the fused launch
            import warnings

            warnings.warn(
                "Simulator(mutation=p): the flip semantics (key split order, one uniform per allele of the finished "
                "gamete) restate chromax from memory and are not pinned by any reference-held vector "
                "(INTEGRATION.md section 4 has the script that would pin them)", stacklevel=2)
            self.engine.set_mutation(mutation)
        self.device = self.engine.device
        self.GEBV_model = TraitModel(self, gm.marker_effects)
        if seed is None:
            seed = int(np.random.randint(0, 2**32, dtype=np.uint64))
        self.random_key = rng.key(seed)

    # ---- RNG -----------------------------------------------------------------------------
    def set_seed(self, seed: int) -> None:
        self.random_key = rng.key(seed)

    # ---- populations ---------------------------------------------------------------------
    def load_population(self, file_name: Union[str, Path]) -> DeviceArray:
        """``.npy`` bool (n, M, 2) (or whitespace text of n rows x 2M values) -> resident genotypes."""
        file_name = Path(file_name)
        if file_name.suffix == ".npy":
            pop = np.load(file_name)
        else:
            pop = np.loadtxt(file_name, dtype=np.uint8)
            pop = pop.reshape(pop.shape[0], -1, 2)
        return self.device_put(pop.astype(np.bool_, copy=False))

    def save_population(self, population, file_name: Union[str, Path]) -> None:
        rows, lead = self._rows_of(population)
        np.save(file_name, self.engine.download_population(rows).reshape(*lead, self.n_markers, 2))

    def device_put(self, population) -> DeviceArray:
        """numpy / torch genotypes (n, M, 2) -> device-resident (n, M, 2) bool view."""
        rows, lead = self._rows_of(population)
        return self.engine.as_genotypes(rows, *lead)

    def _rows_of(self, population) -> Tuple[torch.Tensor, Tuple[int, ...]]:
        """Any genotype array (..., M, 2) -> ((rows, pitch) uint8 device tensor, leading shape).

        Zero-copy when `population` already is a view made by ``as_genotypes``."""
        eng = self.engine
        M, pitch = self.n_markers, eng.pitch
        if isinstance(population, torch.Tensor) and population.device == eng.device:
            p = population.as_subclass(torch.Tensor)
            if p.dim() < 3 or p.shape[-2:] != (M, 2):
                raise ValueError(f"population must have shape (..., {M}, 2), got {tuple(p.shape)}")
            lead = tuple(p.shape[:-2])
            rows = int(np.prod(lead)) if lead else 1
            want, acc = [], pitch
            for d in reversed(lead):
                want.append(acc)
                acc *= d
            want = tuple(reversed(want)) + (2, 1)
            ok = p.element_size() == 1 and all(
                s == w or d == 1 for s, w, d in zip(p.stride(), want, p.shape)
            )
            if ok:
                u8 = p.view(torch.uint8)
                need = u8.storage_offset() + rows * pitch
                if need <= u8.untyped_storage().nbytes():
                    return torch.as_strided(u8, (rows, pitch), (pitch, 1), u8.storage_offset()), lead
            return eng.pack_device(p.reshape(rows, M, 2)), lead
        a = population.detach().cpu().numpy() if isinstance(population, torch.Tensor) else np.asarray(population)
        if a.ndim < 3 or a.shape[-2:] != (M, 2):
            raise ValueError(f"population must have shape (..., {M}, 2), got {a.shape}")
        lead = tuple(a.shape[:-2])
        return eng.upload_population(a.reshape(-1, M, 2)), lead

    # ---- the hot path --------------------------------------------------------------------
    def cross_indexed(self, pool_rows_t, n_envs, pool_rows, idx, n_out, out_rows_t, gebv=None,
                      rng_mode: int = _lib.RNG_SHARED_ENVS, reward=None, reward_next=None) -> None:
        """Fused ``population[actions]`` + ``cross`` + ``GEBV_model`` (+ the per-env max reward);
        advances ``random_key`` once."""
        key = np.ascontiguousarray(self.random_key, np.uint32)
        if reward_next is None:
            self.engine.cross(pool_rows_t, n_envs, pool_rows, idx, n_out, key, rng_mode, out_rows_t, gebv, reward)
        else:
            self.engine.cross(pool_rows_t, n_envs, pool_rows, idx, n_out, key, rng_mode, out_rows_t, gebv, reward,
                              reward_next)
        self.random_key = key

    def cross_indexed_packed(self, pool_packed, n_envs, pool_rows, idx, n_out, out_packed, out_bytes=None, gebv=None,
                             rng_mode: int = _lib.RNG_SHARED_ENVS, reward=None, reward_next=None) -> None:
        """``cross_indexed`` on bit-plane rows; ``out_bytes`` also receives the byte-layout offspring."""
        key = np.ascontiguousarray(self.random_key, np.uint32)
        if reward_next is None:
            self.engine.cross_packed(pool_packed, n_envs, pool_rows, idx, n_out, key, rng_mode, out_packed, out_bytes,
                                     gebv, reward)
        else:
            self.engine.cross_packed(pool_packed, n_envs, pool_rows, idx, n_out, key, rng_mode, out_packed, out_bytes,
                                     gebv, reward, reward_next)
        self.random_key = key

    def cross(self, parents) -> DeviceArray:
        """``Simulator.cross(parents)``: parents (n, 2, M, 2) -> offspring (n, M, 2)."""
        shape = tuple(parents.shape)
        if len(shape) != 4 or shape[1] != 2 or shape[2:] != (self.n_markers, 2):
            raise ValueError(f"parents must have shape (n, 2, {self.n_markers}, 2), got {shape}")
        n = shape[0]
        pool, _ = self._rows_of(parents.reshape(2 * n, self.n_markers, 2))
        idx = torch.arange(2 * n, dtype=torch.int32, device=self.engine.device).reshape(1, n, 2)
        out = self.engine.empty_population(n)
        self.cross_indexed(pool, 1, 2 * n, idx, n, out)
        return self.engine.as_genotypes(out)

    def double_haploid(self, population, n_offspring: int = 1, *, rng_mode: int = _lib.RNG_SHARED_ENVS) -> DeviceArray:
        """``Simulator.double_haploid`` (breeding_programs_env.py:39-41): every individual gives
        `n_offspring` doubled-haploid lines -- one meiosis, the gamete on both strands.

        (..., n, M, 2) -> (..., n, n_offspring, M, 2); leading axes are envs that share the keys
        unless ``rng_mode`` says otherwise.  [EXTERNAL/UNVERIFIED] key plumbing: keys =
        split(key, n * n_offspring + 1); key <- keys[0]; line (i, j) uses keys[1 + i * n_offspring + j]."""
        rows, lead = self._rows_of(population)
        n = lead[-1]
        E = int(np.prod(lead[:-1])) if len(lead) > 1 else 1
        eng = self.engine
        one = torch.arange(n, dtype=torch.int32, device=eng.device).repeat_interleave(n_offspring)
        idx = one.repeat(E).reshape(E, n * n_offspring, 1).expand(-1, -1, 2).contiguous()
        out = eng.empty_population(E * n * n_offspring)
        key = np.ascontiguousarray(self.random_key, np.uint32)
        eng.cross(rows, E, n, idx, n * n_offspring, key, rng_mode | _lib.RNG_DOUBLE_HAPLOID, out, None)
        self.random_key = key
        return eng.as_genotypes(out, *lead[:-1], n, n_offspring)

    def random_crosses(self, population, n_crosses: int, n_offspring: int = 1):
        """n_crosses random pairs without replacement of pairs; each crossed n_offspring times."""
        rows, lead = self._rows_of(population)
        n = rows.shape[0]
        keys = rng.split(self.random_key, 2, self.threefry_layout)
        self.random_key = keys[0]
        all_pairs = self._diallel_indices(np.arange(n))
        pick = rng.choice_no_replace(keys[1], len(all_pairs), n_crosses, self.threefry_layout)
        pairs = np.repeat(np.asarray(all_pairs)[pick], n_offspring, axis=0).astype(np.int32)
        idx = self.engine.to_device(pairs, torch.int32).reshape(1, -1, 2)
        out = self.engine.empty_population(len(pairs))
        self.cross_indexed(rows, 1, n, idx, len(pairs), out)
        return self.engine.as_genotypes(out), pairs

    # ---- breeding values & selection --------------------------------------------------------
    def GEBV(self, population) -> pd.DataFrame:
        """GEBV per individual as a DataFrame with one column per trait (breedgym.py:225-235)."""
        g = self.GEBV_model(population)
        return pd.DataFrame(np.asarray(g), columns=self.trait_names)

    def select(self, population, k: int, f_index: Optional[Callable] = None):
        """The k individuals with the largest index, best first; returns (population[best], best)."""
        if f_index is None:
            f_index = conventional_index(self.GEBV_model)
        scores = f_index(population)
        if not isinstance(scores, torch.Tensor):
            scores = torch.as_tensor(np.asarray(scores, np.float32))
        rows, lead = self._rows_of(population)
        # (n, M, 2): the k best of the population.  (..., n, M, 2): the k best of every group of n
        # (what `vmap(simulator.select)` over lines or envs gives, scripts/time_wheat.py:23-24)
        n = lead[-1]
        groups = int(np.prod(lead[:-1])) if len(lead) > 1 else 1
        best = self.engine.topk(scores.as_subclass(torch.Tensor).reshape(groups, n), k)  # (groups, k)
        flat = (best.to(torch.int64) + torch.arange(groups, device=best.device).unsqueeze(1) * n).to(torch.int32)
        out = self.engine.empty_population(groups * k)
        self.engine.gather(rows, flat.reshape(-1).contiguous(), out)
        return self.engine.as_genotypes(out, *lead[:-1], k), DeviceArray.wrap(best.reshape(*lead[:-1], k))

    @staticmethod
    def _diallel_indices(indices):
        """All unordered pairs (i < j) of `indices`, row-major upper triangle (vec_wrappers.py:66)."""
        is_t = isinstance(indices, torch.Tensor)
        ind = indices.as_subclass(torch.Tensor) if is_t else np.asarray(indices)
        n = len(ind)
        if is_t:
            iu = torch.triu_indices(n, n, offset=1, device=ind.device)
            return torch.stack((ind[iu[0]], ind[iu[1]]), dim=1)
        iu = np.triu_indices(n, k=1)
        return np.stack((ind[iu[0]], ind[iu[1]]), axis=1)

    def corrcoef(self, population) -> DeviceArray:
        """Correlation of every individual's allele vector with the population mean (breedgym.py:240)."""
        rows, _ = self._rows_of(population)
        x = rows[:, : 2 * self.n_markers].to(torch.float32)
        mean = x.mean(dim=0, keepdim=True)
        xc = x - x.mean(dim=1, keepdim=True)
        mc = mean - mean.mean()
        num = (xc * mc).sum(dim=1)
        den = xc.norm(dim=1) * mc.norm()
        return DeviceArray.wrap(num / den)


def conventional_index(GEBV_model: TraitModel) -> Callable:
    """Default selection index: the sum of the GEBVs over traits."""

    def conventional_index_f(pop):
        return GEBV_model(pop).as_subclass(torch.Tensor).sum(dim=-1)

    return conventional_index_f
