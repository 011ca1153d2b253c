"""``BreedGym``: the single (non-vector) environment (reference ``breedgym/breedgym.py:23-242``).

Same hot path as the vector env -- gather + cross + GEBV in one fused launch -- with the
reference's single-env contract: the offspring count is ``len(action)`` (the population changes
size), ``info["GEBV"]`` is a pandas DataFrame with one column per trait, the reward is the mean
GEBV at the horizon, ``GEBV``/``corrcoef`` are cached until the population changes.
Rendering (matplotlib box plots, breedgym.py:159-210) is outside the accelerated path.
"""
from pathlib import Path
from typing import Optional, Tuple, Union

import numpy as np
import pandas as pd
import torch

from . import spaces
from .engine import DeviceArray
from .simulator import Simulator
from .utils.paths import DATA_PATH

GENOME_FILE = DATA_PATH.joinpath("small_geno.npy")


class BreedGym(spaces.Env):
    metadata = {"render_modes": [], "render_fps": 1}

    def __init__(
        self,
        initial_population: Union[str, Path, np.ndarray, torch.Tensor] = GENOME_FILE,
        num_generations: int = 10,
        reward_shaping: bool = False,
        render_mode: Optional[str] = None,
        render_kwargs: Optional[dict] = None,
        **kwargs,
    ):
        if render_mode is not None:
            raise NotImplementedError("rendering is not part of the accelerated hot path")
        self.simulator = Simulator(**kwargs)
        if isinstance(initial_population, (str, Path)):
            if Path(initial_population) == GENOME_FILE and not GENOME_FILE.exists():
                raise FileNotFoundError(
                    f"{GENOME_FILE}: the reference's default population (a 370 x 10k blob its repository strips, "
                    ".MISSING_LARGE_BLOBS) is not shipped here either -- pass initial_population= (a .npy path or an "
                    "(n, M, 2) array; breedgym_amd.synthetic.synthetic_genotypes makes one of the same statistics)"
                )
            germplasm = self.simulator.load_population(initial_population)
        else:
            germplasm = self.simulator.device_put(initial_population)
        self.device = self.simulator.device
        self.germplasm = germplasm
        self._germ_rows, _ = self.simulator._rows_of(germplasm)

        self.num_generations = num_generations
        self.reward_shaping = reward_shaping
        self.render_mode = None

        self._population, self._rows = None, None
        self._GEBV, self._GEBV_cache, self._gebv_dev = None, False, None
        self._corrcoef, self._corrcoef_cache = None, False
        self._set_spaces(tuple(self.germplasm.shape))

        self.step_idx = None
        self.episode_idx = -1

    def _set_spaces(self, shape):
        # the reference builds both spaces anew every step (breedgym.py:99-107); they only change with the shape of
        # the population, and building them is 48 us of a 16-cross step's 144
        if getattr(self, "_spaces_shape", None) == shape:
            return
        n = shape[0]
        self.observation_space = spaces.Box(0, 1, shape=shape, dtype=np.bool_)
        self.action_space = spaces.Sequence(spaces.Tuple((spaces.Discrete(n), spaces.Discrete(n))))
        self._spaces_shape = shape

    def _update_spaces(self):
        self._set_spaces(tuple(self.population.shape))

    # ---- population and its cached statistics (breedgym.py:215-242) ---------------------------
    @property
    def population(self):
        return self._population

    @population.setter
    def population(self, new_pop):
        rows, lead = self.simulator._rows_of(new_pop)
        self._adopt(rows)

    def _adopt(self, rows, gebv_dev=None):
        self._rows = rows
        self._population = self.simulator.engine.as_genotypes(rows)
        self._GEBV_cache, self._gebv_dev = False, gebv_dev
        self._corrcoef_cache = False

    @property
    def GEBV(self) -> pd.DataFrame:
        if not self._GEBV_cache:
            g = self._gebv_dev if self._gebv_dev is not None else self.simulator.engine.gebv(self._rows)
            values = g.cpu().numpy()
            # index objects made once: pd.DataFrame(values, columns=[...]) spends 30 of its 37 us building them
            if getattr(self, "_gebv_columns", None) is None:
                self._gebv_columns = pd.Index(list(self.simulator.trait_names))
            if getattr(self, "_gebv_index", None) is None or len(self._gebv_index) != len(values):
                self._gebv_index = pd.RangeIndex(len(values))
            self._GEBV = pd.DataFrame(values, index=self._gebv_index, columns=self._gebv_columns, copy=False)
            self._GEBV_cache = True
        return self._GEBV

    @property
    def corrcoef(self):
        if not self._corrcoef_cache:
            self._corrcoef = self.simulator.corrcoef(self.population)
            self._corrcoef_cache = True
        return self._corrcoef

    def _get_info(self):
        return {"GEBV": self.GEBV}

    # ---- gymnasium surface ----------------------------------------------------------------------
    def reset(self, seed: Optional[int] = None, options: Optional[dict] = None) -> Tuple[DeviceArray, dict]:
        super().reset(seed=seed)
        if seed is not None:
            self.simulator.set_seed(seed=seed)

        self.step_idx = 0
        self.episode_idx += 1
        eng = self.simulator.engine
        if options is not None and "n_individuals" in options.keys():
            selected = self.np_random.choice(len(self.germplasm), options["n_individuals"], replace=False)
            idx = torch.from_numpy(np.ascontiguousarray(selected, np.int32)).to(eng.device)
            rows = eng.empty_population(len(selected))
            gebv = eng.empty((len(selected), eng.n_traits), torch.float32)
            eng.gather(self._germ_rows, idx, rows, gebv)
            self._adopt(rows, gebv)
        else:
            self._adopt(self._germ_rows)

        self._update_spaces()
        return self.population, self._get_info()

    def step(self, action) -> Tuple[DeviceArray, float, bool, bool, dict]:
        """`action` is (n, 2): each row a couple of parent indices (breedgym.py:136-157)."""
        eng = self.simulator.engine
        a = action.as_subclass(torch.Tensor) if isinstance(action, torch.Tensor) else torch.from_numpy(
            np.ascontiguousarray(np.asarray(action), np.int32)
        )
        idx = a.to(device=eng.device, dtype=torch.int32).reshape(1, -1, 2).contiguous()
        n_out = idx.shape[1]
        out = eng.empty_population(n_out)
        gebv = eng.empty((n_out, eng.n_traits), torch.float32)
        self.simulator.cross_indexed(self._rows, 1, self._rows.shape[0], idx, n_out, out, gebv)
        self._adopt(out, gebv)
        self.step_idx += 1
        self._update_spaces()

        info = self._get_info()
        truncated = self.step_idx == self.num_generations
        if self.reward_shaping or truncated:
            reward = np.mean(self.GEBV.to_numpy())
        else:
            reward = 0
        return self.population, reward, False, truncated, info

    def render(self, file_name=None):
        raise NotImplementedError("rendering is not part of the accelerated hot path")
