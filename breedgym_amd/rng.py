"""Host-side JAX-threefry key chains (jax.random.key / split / permutation), via libbreedhip.

Keys are ``numpy.uint32[2]`` arrays, the raw key data of a ``jax.random.key``.  These calls are
pure host arithmetic: they work without a GPU.
"""
import ctypes

import numpy as np

from . import _lib

_u32p = ctypes.POINTER(ctypes.c_uint32)
_i32p = ctypes.POINTER(ctypes.c_int32)


def _k(key) -> np.ndarray:
    k = np.ascontiguousarray(np.asarray(key, dtype=np.uint32).reshape(2))
    return k


def key(seed: int) -> np.ndarray:
    """``jax.random.key(seed)`` (x64 disabled): (0, seed mod 2**32)."""
    out = np.zeros(2, np.uint32)
    _lib.lib().bh_key_from_seed(ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), out.ctypes.data_as(_u32p))
    return out


def split(k, num: int = 2, layout="original") -> np.ndarray:
    """``jax.random.split(key, num)`` -> (num, 2) uint32.  ``layout``: "original" (jax_threefry_partitionable off)
    or "partitionable" (the default of jax >= 0.5)."""
    k = _k(k)
    out = np.empty((int(num), 2), np.uint32)
    _lib.check(_lib.lib().bh_key_split(k.ctypes.data_as(_u32p), int(num), _lib.layout_code(layout),
                                       out.ctypes.data_as(_u32p)))
    return out


def permutation(k, n: int, layout="original") -> np.ndarray:
    """``jax.random.permutation(key, n)`` -> (n,) int32."""
    k = _k(k)
    out = np.empty(max(int(n), 1), np.int32)
    _lib.check(_lib.lib().bh_permutation(k.ctypes.data_as(_u32p), int(n), _lib.layout_code(layout),
                                         out.ctypes.data_as(_i32p)))
    return out[: int(n)]


def choice_no_replace(k, n_inputs: int, n_draws: int, layout="original") -> np.ndarray:
    """``jax.random.choice(key, n_inputs, (n_draws,), replace=False)`` = permutation[:n_draws]."""
    if n_draws > n_inputs:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    return permutation(k, n_inputs, layout)[:n_draws]


def founders(env_key, n_envs: int, n_germplasm: int, n: int, layout="original"):
    """Founder draw of ``VecBreedGym.reset`` (vec_env.py:120-128) -> ((n_envs, n) int32, new_key)."""
    k = _k(env_key).copy()
    out = np.empty((int(n_envs), int(n)), np.int32)
    _lib.check(
        _lib.lib().bh_founders(k.ctypes.data_as(_u32p), int(n_envs), int(n_germplasm), int(n),
                               _lib.layout_code(layout), out.ctypes.data_as(_i32p))
    )
    return out, k
