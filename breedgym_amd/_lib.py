"""ctypes binding of libbreedhip.so (the C ABI in include/breedhip.h).

The library is HIP-only.  If it has not been built, or no gfx950 device is visible, the
compute entry points raise -- there is no CPU fallback anywhere in this package.
"""
import ctypes
import os
import shutil
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent
_ROOT = _PKG.parent
LIB_PATH = _PKG / "libbreedhip.so"
_CSRC = _PKG / "csrc"
_UNITS = ("breedhip", "cross", "gebv", "select")  # translation units (csrc/ctx.h says what lives where)
_BUILD = _CSRC / "_build"


def _sources():
    return sorted(_CSRC.glob("*.hip")) + sorted(_CSRC.glob("*.h")) + [_ROOT / "include" / "breedhip.h"]

BH_OK = 0
BH_ERR_BAD_ARG = -1
BH_ERR_OOM = -2
BH_ERR_HIP = -3
BH_ERR_UNSUPPORTED = -4
BH_ERR_NO_DEVICE = -5
BH_ERR_BAD_GENOTYPE = -6

RNG_SHARED_ENVS = 0
RNG_INDEPENDENT = 1
RNG_DOUBLE_HAPLOID = 2  # flag: one gamete per row, written to both allele slots

THREEFRY_LAYOUTS = {"original": 0, "partitionable": 1}  # bh_threefry_layout


def layout_code(layout) -> int:
    if isinstance(layout, str):
        if layout not in THREEFRY_LAYOUTS:
            raise ValueError(f"threefry_layout must be one of {sorted(THREEFRY_LAYOUTS)}, got {layout!r}")
        return THREEFRY_LAYOUTS[layout]
    if layout not in (0, 1):
        raise ValueError(f"threefry_layout must be 0 or 1, got {layout!r}")
    return int(layout)


class BreedHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libbreedhip error {code}: {msg}")
        self.code = code


def build_native(force: bool = False, verbose: bool = False, extra_flags=()) -> Path:
    """Compile libbreedhip.so for gfx950 with hipcc (cross-compiles without a GPU): the translation units in
    parallel, then one link."""
    if not force and LIB_PATH.exists():
        newest = max(p.stat().st_mtime for p in _sources())
        if LIB_PATH.stat().st_mtime >= newest:
            return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libbreedhip.so")
    _BUILD.mkdir(exist_ok=True)
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", f"-I{_ROOT / 'include'}", *extra_flags]
    jobs = []
    for u in _UNITS:
        cmd = [hipcc, *flags, "-c", str(_CSRC / f"{u}.hip"), "-o", str(_BUILD / f"{u}.o")]
        if verbose:
            print(" ".join(cmd))
        jobs.append((u, subprocess.Popen(cmd)))
    failed = [u for u, j in jobs if j.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed on {failed}")
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH)] + [str(_BUILD / f"{u}.o") for u in _UNITS]
    if verbose:
        print(" ".join(link))
    subprocess.run(link, check=True)
    return LIB_PATH


_lib = None

_c = ctypes
_vp, _i64, _int = _c.c_void_p, _c.c_int64, _c.c_int
_u32p = _c.POINTER(_c.c_uint32)
_i32p = _c.POINTER(_c.c_int32)
_f32p = _c.POINTER(_c.c_float)

# name -> (restype, argtypes); kept in step with include/breedhip.h (tests/test_abi.py checks it)
SIGNATURES = {
    "bh_version": (_c.c_char_p, []),
    "bh_last_error": (_c.c_char_p, []),
    "bh_device_count": (_int, [_c.POINTER(_int)]),
    "bh_create": (_int, [_int, _i64, _i64, _f32p, _f32p, _c.POINTER(_vp)]),
    "bh_destroy": (_int, [_vp]),
    "bh_set_stream": (_int, [_vp, _vp]),
    "bh_sync": (_int, [_vp]),
    "bh_clear_error": (_int, [_vp, _u32p]),
    "bh_debug_inject_error": (_int, [_vp, _c.c_uint32, _int]),
    "bh_debug_count_dirty_cells": (_int, [_vp, _c.POINTER(_c.c_int64)]),
    "bh_release_scratch": (_int, [_vp]),
    "bh_row_pitch": (_i64, [_vp]),
    "bh_n_markers": (_i64, [_vp]),
    "bh_n_traits": (_i64, [_vp]),
    "bh_max_markers": (_i64, []),
    "bh_alloc": (_int, [_vp, _i64, _c.POINTER(_vp)]),
    "bh_free": (_int, [_vp, _vp]),
    "bh_memcpy_h2d": (_int, [_vp, _vp, _vp, _i64]),
    "bh_memcpy_d2h": (_int, [_vp, _vp, _vp, _i64]),
    "bh_upload_population": (_int, [_vp, _vp, _i64, _vp, _int]),
    "bh_download_population": (_int, [_vp, _vp, _i64, _vp]),
    "bh_cross": (_int, [_vp, _vp, _i64, _i64, _vp, _i64, _u32p, _int, _vp, _vp, _vp, _vp, _i64]),
    "bh_set_mask_prefetch": (_int, [_vp, _int]),
    "bh_gather": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "bh_gebv": (_int, [_vp, _vp, _i64, _vp]),
    "bh_reward_max": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "bh_topk": (_int, [_vp, _vp, _i64, _i64, _i64, _vp]),
    "bh_selection_actions": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _u32p, _vp]),
    "bh_pair_actions": (_int, [_vp, _vp, _i64, _i64, _i64, _vp]),
    "bh_winner_layout": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "bh_optimal_haploid": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "bh_packed_pitch": (_i64, [_vp]),
    "bh_pack": (_int, [_vp, _vp, _i64, _vp]),
    "bh_cross_packed": (_int, [_vp, _vp, _i64, _i64, _vp, _i64, _u32p, _int, _vp, _vp, _vp, _vp, _vp, _i64]),
    "bh_hint_next_cross": (_int, [_vp, _i64, _i64, _int]),
    "bh_step_bind": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _i64]),
    "bh_step": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _u32p, _int]),
    "bh_gather_packed": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "bh_gather_packed_deferred": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "bh_key_from_seed": (None, [_c.c_uint64, _u32p]),
    "bh_set_threefry_layout": (_int, [_vp, _int]),
    "bh_set_mutation": (_int, [_vp, _c.c_float]),
    "bh_set_env_window": (_int, [_vp, _i64, _i64]),
    "bh_set_cross_window": (_int, [_vp, _i64, _i64]),
    "bh_key_split": (_int, [_u32p, _i64, _int, _u32p]),
    "bh_permutation": (_int, [_u32p, _i64, _int, _i32p]),
    "bh_founders": (_int, [_u32p, _i64, _i64, _i64, _int, _i32p]),
    "bh_founders_device": (_int, [_vp, _u32p, _i64, _i64, _i64, _vp]),
    "bh_founders_deferred": (_int, [_vp, _u32p, _i64, _i64, _i64, _vp]),
}


def lib() -> ctypes.CDLL:
    """Load libbreedhip.so; raises if it has not been built (see __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(needs hipcc).  breedgym_amd has no CPU fallback."
            )
        L = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc == BH_OK:
        return
    msg = lib().bh_last_error().decode("utf-8", "replace")
    if rc in (BH_ERR_BAD_ARG, BH_ERR_BAD_GENOTYPE):
        raise ValueError(f"libbreedhip: {msg}")
    if rc == BH_ERR_OOM:
        raise MemoryError(f"libbreedhip: {msg}")
    raise BreedHipError(rc, msg)


def device_count() -> int:
    n = _int(0)
    rc = lib().bh_device_count(ctypes.byref(n))
    return n.value if rc == BH_OK else 0
