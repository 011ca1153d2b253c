"""ctypes binding of the C oracle (TEST INFRASTRUCTURE -- see oracle/__init__.py)."""
import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "libbreedoracle.so"
_lib = None

c_i64 = ctypes.c_int64
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(force: bool = False) -> Path:
    """Compile breed_oracle.c with gcc (seconds).  Idempotent."""
    src = _HERE / "breed_oracle.c"
    hdr = _HERE / "breed_oracle.h"
    stale = (
        force
        or not _LIB_PATH.exists()
        or _LIB_PATH.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime)
    )
    if stale:
        subprocess.run(["make", "-s", "-C", str(_HERE)], check=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(str(_LIB_PATH))
        L.bo_uniform_f32.restype = ctypes.c_float
        L.bo_uniform_f32.argtypes = [ctypes.c_uint32]
        L.bo_threshold23.restype = ctypes.c_uint32
        L.bo_threshold23.argtypes = [ctypes.c_float]
        L.bo_cross.restype = ctypes.c_int
        L.bo_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a, typ):
    return a.ctypes.data_as(typ)


def _key(key):
    k = np.ascontiguousarray(np.asarray(key, dtype=np.uint32).reshape(2)).copy()
    return k


LAYOUTS = {"original": 0, "partitionable": 1}


def set_threefry_layout(layout) -> None:
    """Which JAX counter layout split / random_bits (and everything on top) follow; a process-wide switch."""
    lib().bo_set_threefry_layout(ctypes.c_int(LAYOUTS[layout] if isinstance(layout, str) else int(layout)))


def threefry_layout() -> str:
    return "partitionable" if lib().bo_threefry_layout() else "original"


def threefry2x32(k0, k1, c0, c1):
    o0, o1 = ctypes.c_uint32(), ctypes.c_uint32()
    lib().bo_threefry2x32(
        ctypes.c_uint32(k0), ctypes.c_uint32(k1), ctypes.c_uint32(c0), ctypes.c_uint32(c1),
        ctypes.byref(o0), ctypes.byref(o1),
    )
    return o0.value, o1.value


def key_from_seed(seed: int) -> np.ndarray:
    k = np.zeros(2, np.uint32)
    lib().bo_key_from_seed(ctypes.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), _p(k, _u32p))
    return k


def split(key, num: int) -> np.ndarray:
    k = _key(key)
    out = np.empty((num, 2), np.uint32)
    lib().bo_split(_p(k, _u32p), c_i64(num), _p(out, _u32p))
    return out


def random_bits(key, n: int) -> np.ndarray:
    k = _key(key)
    out = np.empty(max(n, 1), np.uint32)
    lib().bo_random_bits(_p(k, _u32p), c_i64(n), _p(out, _u32p))
    return out[:n]


def uniform_f32(bits) -> np.ndarray:
    b = np.asarray(bits, np.uint32)
    return np.array([lib().bo_uniform_f32(int(x)) for x in b.ravel()], np.float32).reshape(b.shape)


def threshold23(r) -> np.ndarray:
    r = np.asarray(r, np.float32)
    return np.array([lib().bo_threshold23(float(x)) for x in r.ravel()], np.uint32).reshape(r.shape)


def permutation(key, N: int) -> np.ndarray:
    k = _key(key)
    out = np.empty(max(N, 1), np.int32)
    lib().bo_permutation(_p(k, _u32p), c_i64(N), _p(out, _i32p))
    return out[:N]


def meiosis_mask(rec_key, r) -> np.ndarray:
    k = _key(rec_key)
    r = np.ascontiguousarray(r, np.float32)
    out = np.empty(max(len(r), 1), np.uint8)
    lib().bo_meiosis_mask(_p(k, _u32p), _p(r, _f32p), c_i64(len(r)), _p(out, _u8p))
    return out[: len(r)]


def cross(pool, idx, r, key, rng_mode: int = 0, return_masks: bool = False, mutation: float = 0.0):
    """pool (E,n_pool,M,2) u8/bool; idx (E,n_out,2) -> (out (E,n_out,M,2) u8, new_key[, masks])."""
    pool = np.ascontiguousarray(pool).view(np.uint8) if pool.dtype == np.bool_ else np.ascontiguousarray(pool, np.uint8)
    if pool.ndim == 3:
        pool = pool[None]
    idx = np.ascontiguousarray(idx, np.int32)
    if idx.ndim == 2:
        idx = idx[None]
    E, n_pool, M, _ = pool.shape
    n_out = idx.shape[1]
    assert idx.shape == (E, n_out, 2)
    r = np.ascontiguousarray(r, np.float32)
    assert r.shape == (M,)
    k = _key(key)
    out = np.empty((E, n_out, M, 2), np.uint8)
    masks = np.empty((E, n_out, 2, M), np.uint8) if return_masks else None
    rc = lib().bo_cross_mutation(
        _p(pool, _u8p), c_i64(E), c_i64(n_pool), _p(idx, _i32p), c_i64(n_out), c_i64(M),
        _p(r, _f32p), _p(k, _u32p), ctypes.c_int(rng_mode), ctypes.c_float(mutation), _p(out, _u8p),
        _p(masks, _u8p) if return_masks else None,
    )
    if rc != 0:
        raise IndexError("empty parent pool")
    return (out, k, masks) if return_masks else (out, k)


def gebv(pop, effects, want64: bool = False):
    """pop (..., M, 2) u8/bool, effects (M,T) f32 -> (..., T) f32 [and f64]."""
    pop = np.ascontiguousarray(pop).view(np.uint8) if pop.dtype == np.bool_ else np.ascontiguousarray(pop, np.uint8)
    lead = pop.shape[:-2]
    M = pop.shape[-2]
    eff = np.ascontiguousarray(effects, np.float32)
    if eff.ndim == 1:
        eff = eff[:, None]
    T = eff.shape[1]
    rows = int(np.prod(lead)) if lead else 1
    o32 = np.empty((rows, T), np.float32)
    o64 = np.empty((rows, T), np.float64)
    lib().bo_gebv(_p(pop, _u8p), c_i64(rows), c_i64(M), _p(eff, _f32p), c_i64(T), _p(o32, _f32p), _p(o64, _f64p))
    o32 = o32.reshape(*lead, T)
    return (o32, o64.reshape(*lead, T)) if want64 else o32


def reward_max(gebv_arr) -> np.ndarray:
    g = np.ascontiguousarray(gebv_arr, np.float32)
    E = g.shape[0]
    out = np.empty(E, np.float32)
    lib().bo_reward_max(_p(g, _f32p), c_i64(E), c_i64(g.size // max(E, 1)), _p(out, _f32p))
    return out


def founders(env_key, E: int, N: int, n: int):
    k = _key(env_key)
    out = np.empty((E, n), np.int32)
    lib().bo_founders(_p(k, _u32p), c_i64(E), c_i64(N), c_i64(n), _p(out, _i32p))
    return out, k


def topk(scores, k: int) -> np.ndarray:
    s = np.ascontiguousarray(scores, np.float32)
    if not 0 <= k <= len(s):
        raise ValueError("topk: need 0 <= k <= n")
    out = np.empty(k, np.int32)
    lib().bo_topk(_p(s, _f32p), c_i64(len(s)), c_i64(k), _p(out, _i32p))
    return out


def selection_actions(scores, k: int, n_crosses: int, env_key):
    s = np.ascontiguousarray(scores, np.float32)
    E, n = s.shape
    key = _key(env_key)
    out = np.empty((E, n, 2), np.int32)
    lib().bo_selection_actions(_p(s, _f32p), c_i64(E), c_i64(n), c_i64(k), c_i64(n_crosses), _p(key, _u32p), _p(out, _i32p))
    return out, key


def optimal_haploid(pop, effects, block_start=None):
    """-> (OHP (n, M) u8, OHV (n, T) f32); block_start: marker offsets of the blocks, or None."""
    g = np.ascontiguousarray(pop, np.uint8)
    eff = np.ascontiguousarray(effects, np.float32)
    eff = eff if eff.ndim == 2 else eff[:, None]
    n, M, _ = g.shape
    T = eff.shape[1]
    bs = None if block_start is None else np.ascontiguousarray(block_start, np.int32)
    ohp, ohv = np.empty((n, M), np.uint8), np.empty((n, T), np.float32)
    lib().bo_optimal_haploid(_p(g, _u8p), c_i64(n), c_i64(M), _p(eff, _f32p), c_i64(T),
                             None if bs is None else _p(bs, _i32p), c_i64(0 if bs is None else len(bs) - 1),
                             _p(ohp, _u8p), _p(ohv, _f32p))
    return ohp, ohv


def num_threads() -> int:
    return lib().bo_num_threads()


def set_num_threads(n: int) -> None:
    lib().bo_set_num_threads(ctypes.c_int(n))
    os.environ["OMP_NUM_THREADS"] = str(n)
