"""Action-shaping wrappers: policy scores -> parent index pairs, kept on the device.

Drop-ins for ``breedgym/vector/vec_wrappers.py`` (reference lines cited per class).  They sit
directly in front of ``VecBreedGym.step``; the conversion runs on the GPU so that the scores
(typically the previous step's ``infos["GEBV"]``) never visit the host:

* ``SelectionScores`` -- one fused launch per step (``bh_selection_actions``): top-k, diallel
  pairs, random subset without replacement, repeat.
* ``PairScores``      -- top-n of the n x n pair scores with the top-k kernel, then a handful of
  small device ops for the softmax-proportional repeat.
* ``RavelIndex``      -- flat index -> (row, col).
"""
from math import prod
from typing import Optional

import numpy as np
import torch

from .. import spaces
from ..engine import DeviceArray
from .vec_env import VecBreedGym


class VectorWrapper:
    """Minimal stand-in for gymnasium's VectorWrapper: forwards unknown attributes to the env."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return getattr(self.env, "unwrapped", self.env)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, actions):
        return self.env.step(actions)

    def close(self, **kwargs):
        return self.env.close(**kwargs)


def _as_device_f32(x, engine) -> torch.Tensor:
    t = x.as_subclass(torch.Tensor) if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    return t.to(device=engine.device, dtype=torch.float32).contiguous()


class SelectionScores(VectorWrapper):
    """One score per individual; the k best are crossed at random (vec_wrappers.py:15-85)."""

    def __init__(self, vec_env: Optional[VecBreedGym] = None, k: Optional[int] = None,
                 n_crosses: Optional[int] = None, **kwargs):
        if vec_env is None:
            vec_env = VecBreedGym(**kwargs)
        super().__init__(vec_env)

        # same rules and defaults as vec_wrappers.py:27-51 (a tenth of the generation; every pair of the half
        # diallel among them), same exception type; the messages are this package's own
        n = self.individual_per_gen
        self.k = n // 10 if k is None else int(k)
        if self.k > n:
            raise ValueError(f"k={self.k} parents cannot be chosen among the {n} individuals of a generation")
        pairs = self.k * (self.k - 1) // 2
        self.n_crosses = pairs if n_crosses is None else int(n_crosses)
        if self.n_crosses > pairs:
            raise ValueError(f"n_crosses={self.n_crosses}, but k={self.k} parents form only {pairs} distinct pairs")
        if self.n_crosses > n:
            raise ValueError(f"n_crosses={self.n_crosses} crosses do not fit the {n} offspring slots of a generation")
        if self.n_crosses < 1:
            raise ValueError(f"k={self.k} parents leave no pair to cross")

        self.single_action_space = spaces.Box(-1e5, 1e5, shape=(self.individual_per_gen,))
        self.action_space = spaces.Box(-1e5, 1e5, shape=(self.num_envs, self.individual_per_gen))

    def _convert_actions(self, actions):
        """(E, n) scores -> (E, n, 2) parent indices; advances the env's ``random_key``."""
        eng = self.simulator.engine
        s = _as_device_f32(actions, eng).reshape(self.num_envs, -1)
        low, self.env.random_key = eng.selection_actions(s, self.k, self.n_crosses, self.env.random_key)
        return low

    def step(self, actions):
        return self.env.step(self._convert_actions(actions))


class PairScores(VectorWrapper):
    """One score per ordered pair; the n best pairs get offspring in proportion to
    softmax(score) (vec_wrappers.py:88-126)."""

    def __init__(self, vec_env: Optional[VecBreedGym] = None, **kwargs):
        if vec_env is None:
            vec_env = VecBreedGym(**kwargs)
        super().__init__(vec_env)
        self.n_crosses = self.individual_per_gen
        action_shape = self.n_crosses, self.n_crosses
        self.single_action_space = spaces.Box(-1e5, 1e5, shape=action_shape)
        self.action_space = spaces.Box(-1e5, 1e5, shape=(self.num_envs, *action_shape))

    def _convert_actions(self, actions):
        eng = self.simulator.engine
        n = self.n_crosses
        if hasattr(eng, "pair_actions") and n <= 16384:  # one fused kernel behind the top-k (bh_pair_actions)
            return eng.pair_actions(_as_device_f32(actions, eng), n)
        # the same in tensor operations (engines without the kernel: the CPU test double)
        s = _as_device_f32(actions, eng).reshape(self.num_envs, -1)  # action.flatten(), whatever its shape (vec_wrappers.py:102)
        best = eng.topk(s, n).to(torch.int64)  # (E, n) flat indices, best first
        vals = torch.gather(s, 1, best)
        # jax.nn.softmax in f32, spelled out so that the one order-dependent step -- the sum -- is done
        # in f64 and rounded once: ceil() makes the offspring counts sensitive to the last bit
        ex = torch.exp(vals - vals.max(dim=1, keepdim=True).values)
        sm = ex / ex.sum(dim=1, keepdim=True, dtype=torch.float64).to(torch.float32)
        per_cross = torch.ceil(sm * n).to(torch.int64)  # offspring per pair
        pairs = torch.stack((torch.div(best, n, rounding_mode="floor"), best % n), dim=2)
        # jnp.repeat(pairs, per_cross, total_repeat_length=n): slot s takes the first pair whose
        # cumulative count exceeds s
        ends = torch.cumsum(per_cross, dim=1)
        slots = torch.arange(n, device=eng.device).expand(self.num_envs, n).contiguous()
        which = torch.searchsorted(ends, slots, right=True).clamp_(max=n - 1)
        return torch.gather(pairs, 1, which.unsqueeze(-1).expand(-1, -1, 2)).to(torch.int32).contiguous()

    def step(self, actions):
        low = self._convert_actions(actions)
        obs, rew, ter, tru, infos = self.env.step(low)
        infos["low_level_actions"] = DeviceArray.wrap(low)
        return obs, rew, ter, tru, infos


class RavelIndex(VectorWrapper):
    """Flat indices into the wrapped action shape -> index pairs (vec_wrappers.py:129-155)."""

    def __init__(self, vec_env):
        super().__init__(vec_env)
        self.action_shape = tuple(self.env.single_action_space.shape)
        n_elems = prod(self.action_shape)
        n_vec = np.full((self.individual_per_gen,), n_elems)
        self.single_action_space = spaces.MultiDiscrete(n_vec)
        self.action_space = spaces.MultiDiscrete(np.broadcast_to(n_vec[None, ...], (self.num_envs, *n_vec.shape)))

    def _convert_actions(self, actions):
        eng = self.simulator.engine
        a = actions.as_subclass(torch.Tensor) if isinstance(actions, torch.Tensor) else torch.as_tensor(np.asarray(actions))
        a = a.to(device=eng.device, dtype=torch.int64)
        idx = []
        for d in reversed(self.action_shape):
            idx.append(a % d)
            a = torch.div(a, d, rounding_mode="floor")
        return torch.stack(list(reversed(idx)), dim=-1).to(torch.int32).contiguous()

    def step(self, actions):
        return self.env.step(self._convert_actions(actions))
