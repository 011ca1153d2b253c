"""Gymnasium spaces, or a minimal stand-in when gymnasium is not installed.

The environments only *describe* their interface with spaces (vec_env.py:57-68,
breedgym.py:47-58); nothing on the hot path depends on them.  If gymnasium is importable its
real classes are used so that the environments are genuine ``gymnasium`` envs.
"""
import numpy as np

try:  # pragma: no cover - depends on the installation
    import gymnasium as gym
    from gymnasium import spaces as _spaces

    Box, Discrete, Tuple, Sequence, Dict, MultiDiscrete = (
        _spaces.Box, _spaces.Discrete, _spaces.Tuple, _spaces.Sequence, _spaces.Dict, _spaces.MultiDiscrete,
    )
    Env = gym.Env
    Wrapper = gym.Wrapper
    HAVE_GYMNASIUM = True
except ImportError:
    HAVE_GYMNASIUM = False

    class _Space:
        def __init__(self, shape=None, dtype=None, seed=None):
            self.shape = None if shape is None else tuple(shape)
            self.dtype = None if dtype is None else np.dtype(dtype)
            self._rng = np.random.default_rng(seed)

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            super().__init__(shape, dtype, seed)
            self.low, self.high = low, high

        def sample(self):
            if np.issubdtype(self.dtype, np.floating):
                return self._rng.uniform(self.low, self.high, self.shape).astype(self.dtype)
            if self.dtype == np.bool_:
                return self._rng.integers(0, 2, self.shape).astype(np.bool_)
            return self._rng.integers(self.low, np.asarray(self.high) + 1, self.shape).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"

    class Discrete(_Space):
        def __init__(self, n, start=0, seed=None):
            super().__init__((), np.int64, seed)
            self.n, self.start = int(n), int(start)

        def sample(self):
            return int(self._rng.integers(self.start, self.start + self.n))

        def contains(self, x):
            return self.start <= int(x) < self.start + self.n

        def __repr__(self):
            return f"Discrete({self.n}, start={self.start})"

    class MultiDiscrete(_Space):
        def __init__(self, nvec, dtype=np.int64, seed=None):
            self.nvec = np.asarray(nvec, dtype=dtype)
            super().__init__(self.nvec.shape, dtype, seed)

        def sample(self):
            return (self._rng.random(self.nvec.shape) * self.nvec).astype(self.dtype)

    class Tuple(_Space):
        def __init__(self, spaces, seed=None):
            super().__init__(None, None, seed)
            self.spaces = tuple(spaces)

        def sample(self):
            return tuple(s.sample() for s in self.spaces)

    class Sequence(_Space):
        def __init__(self, space, seed=None):
            super().__init__(None, None, seed)
            self.feature_space = space

        def sample(self):
            n = int(self._rng.geometric(0.25))
            return tuple(self.feature_space.sample() for _ in range(n))

    class Dict(_Space):
        def __init__(self, spaces, seed=None):
            super().__init__(None, None, seed)
            self.spaces = dict(spaces)

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

    class Env:
        """The sliver of gymnasium.Env the environments rely on (np_random seeding)."""

        metadata = {}
        np_random = None

        def reset(self, *, seed=None, options=None):
            if seed is not None or self.np_random is None:
                self.np_random = np.random.default_rng(seed)

        @property
        def unwrapped(self):
            return self

        def close(self):
            pass

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            if name.startswith("_"):
                raise AttributeError(name)
            return getattr(self.env, name)

        @property
        def unwrapped(self):
            return self.env.unwrapped

        @property
        def np_random(self):
            return self.env.np_random

        def reset(self, **kwargs):
            return self.env.reset(**kwargs)

        def step(self, action):
            return self.env.step(action)
