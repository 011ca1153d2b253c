"""CPU: the C-ABI library loads, exports every symbol include/breedhip.h declares, its ctypes
table matches the header, and its host-side RNG plumbing equals the oracle.  No GPU compute."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

from breedgym_amd import _lib
from breedgym_amd import rng as hrng
from oracle import c_oracle as co

ROOT = Path(__file__).resolve().parents[1]
HEADER = ROOT / "include" / "breedhip.h"


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(bh_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(native_lib):
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(native_lib, s), f"{s} declared in breedhip.h but not exported"


def test_ctypes_table_matches_header(native_lib):
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_version_and_errors(native_lib):
    assert b"gfx950" in native_lib.bh_version()
    import breedgym_amd

    assert breedgym_amd.__version__.encode() in native_lib.bh_version()  # one version: the package's is the library's
    # argument validation happens before any device work, so it is testable without a GPU
    out = ctypes.c_void_p()
    r = np.zeros(4, np.float32)
    rc = native_lib.bh_create(0, 0, 1, r.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                              r.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.byref(out))
    assert rc == _lib.BH_ERR_BAD_ARG and b"n_markers" in native_lib.bh_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)
    rc = native_lib.bh_create(0, native_lib.bh_max_markers() + 1, 1,
                              r.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                              r.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.byref(out))
    assert rc == _lib.BH_ERR_UNSUPPORTED
    assert native_lib.bh_row_pitch(None) == 0


def test_no_device_fails_loudly(native_lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from breedgym_amd import HipEngine

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HipEngine(np.full(8, 0.1, np.float32), np.ones((8, 1), np.float32))


def test_product_never_imports_oracle():
    root = Path(__file__).resolve().parents[1]
    for f in list((root / "breedgym_amd").rglob("*.py")) + list((root / "breedgym_amd").rglob("*.h*")):
        txt = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
        assert "breed_oracle" not in txt, f


def test_host_rng_equals_oracle(native_lib):
    for seed in (0, 7, 2**31 + 5, 2**32 - 1, 2**40 + 3):
        assert np.array_equal(hrng.key(seed), co.key_from_seed(seed))
    k = hrng.key(7)
    for num in (1, 2, 3, 17, 401):
        assert np.array_equal(hrng.split(k, num), co.split(k, num))
    for n in (1, 2, 3, 64, 370, 1001, 70000):
        assert np.array_equal(hrng.permutation(k, n), co.permutation(k, n)), n
    f, nk = hrng.founders(k, 5, 370, 200)
    fo, nko = co.founders(k, 5, 370, 200)
    assert np.array_equal(f, fo) and np.array_equal(nk, nko)
    with pytest.raises(ValueError):
        hrng.founders(k, 2, 10, 11)


def test_host_rng_golden(native_lib, golden_dir):
    z = np.load(golden_dir / "golden_host_rng.npz")
    f, k = hrng.founders(hrng.key(7), 4, 370, 200)
    assert np.array_equal(f, z["founders"]) and np.array_equal(k, z["founders_key"])
    assert np.array_equal(hrng.permutation(hrng.key(5), 70000)[:64], z["perm70000_head"])


def test_bench_launcher_is_loud_without_a_gpu():
    """`python bench.py --gpus 2` on a box without a GPU: the launcher says so and exits non-zero (it starts no rank)."""
    import os
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible")
    ROOT = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "needs a GPU" in r.stderr and r.stdout.strip() == ""


def test_gymnasium_registration_path(tmp_path):
    """breedgym_amd/__init__.py registers the reference's six ids (breedgym/__init__.py:4-34) when gymnasium imports.
    gymnasium is not installed on any machine of this pipeline, so the branch never ran: here a stand-in package of that
    name -- `register` keeps a table, `make` resolves "module:attr" entry points the way gymnasium does -- is put on the
    path of a fresh interpreter, and a registered id is made and stepped through it (oracle-backed engine, CPU)."""
    import subprocess
    import sys
    import textwrap

    pkg = tmp_path / "gymnasium"
    (pkg / "envs").mkdir(parents=True)
    (pkg / "__init__.py").write_text(textwrap.dedent('''
        import importlib
        from .envs.registration import registry, register

        def make(id, **kwargs):
            module, attr = registry[id.split(":")[-1]].split(":")
            return getattr(importlib.import_module(module), attr)(**kwargs)
    '''))
    (pkg / "envs" / "__init__.py").write_text("")
    (pkg / "envs" / "registration.py").write_text(textwrap.dedent('''
        registry = {}

        def register(id, entry_point, **kwargs):
            registry[id] = entry_point
    '''))
    prog = textwrap.dedent(f'''
        import sys
        sys.path[:0] = [{str(tmp_path)!r}, {str(ROOT)!r}, {str(ROOT / "tests")!r}]
        import gymnasium as gym
        import numpy as np
        import breedgym_amd
        from oracle_engine import OracleEngine
        from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
        assert set(gym.registry) == set(breedgym_amd.REGISTRY) and len(gym.registry) == 6, gym.registry
        gm, geno = synthetic_genetic_map(64, n_chr=2, seed=1), synthetic_genotypes(20, 64, seed=2)
        eng = lambda: OracleEngine(gm.recombination_vec, gm.marker_effects)
        env = gym.make("VecBreedGym", num_envs=2, initial_population=geno, genetic_map=gm, individual_per_gen=6, engine=eng())
        pop, _ = env.reset(seed=3)
        pop, rew, ter, tru, info = env.step(np.zeros((2, 6, 2), np.int32))
        assert pop.shape == (2, 6, 64, 2) and len(rew) == 2
        twin = breedgym_amd.make("VecBreedGym", num_envs=2, initial_population=geno, genetic_map=gm, individual_per_gen=6, engine=eng())
        twin.reset(seed=3)
        assert np.array_equal(np.asarray(twin.step(np.zeros((2, 6, 2), np.int32))[0]), np.asarray(pop))
        for id in gym.registry:  # every entry point resolves to a class of this package
            module, attr = gym.registry[id].split(":")
            assert module.startswith("breedgym_amd") and isinstance(getattr(__import__(module, fromlist=[attr]), attr), type)
        print("registered", sorted(gym.registry))
    ''')
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "registered ['BreedGym', 'KBestBreedGym', 'PairScores', 'SelectionScores', 'SimplifiedBreedGym', 'VecBreedGym']" in r.stdout
