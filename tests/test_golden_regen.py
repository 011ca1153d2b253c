"""CPU: every fixture under tests/golden/ is reproducible from its committed generator -- running
tests/golden/make_golden.py again changes nothing (the published vectors are typed in there, the .npz
files are inputs + the oracle's outputs for fixed seeds)."""
import importlib.util
import json
from pathlib import Path

import numpy as np
import pytest

GOLDEN = Path(__file__).resolve().parent / "golden"


def _generator():
    spec = importlib.util.spec_from_file_location("make_golden", GOLDEN / "make_golden.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_published_vectors_regenerate_to_the_committed_file(tmp_path):
    mg = _generator()
    mg.OUT = tmp_path
    mg.published()
    assert json.loads((tmp_path / "kat_published.json").read_text()) == json.loads((GOLDEN / "kat_published.json").read_text())
    assert (tmp_path / "kat_published.json").read_bytes() == (GOLDEN / "kat_published.json").read_bytes()


def test_regenerating_changes_nothing(tmp_path):
    mg = _generator()
    if not mg.REF_DATA.exists():
        pytest.skip("the reference's shipped sample data (inputs of two fixtures) is not on this machine")
    mg.main(tmp_path)
    committed = sorted(p.name for p in GOLDEN.iterdir() if p.suffix in (".npz", ".npy", ".json"))
    made = sorted(p.name for p in tmp_path.iterdir())
    assert made == committed, "the generator and the committed fixtures name different files"
    for name in committed:
        a, b = GOLDEN / name, tmp_path / name
        if name.endswith(".npz"):
            x, y = np.load(a), np.load(b)
            assert sorted(x.files) == sorted(y.files), name
            for k in x.files:
                assert x[k].dtype == y[k].dtype and np.array_equal(x[k], y[k]), (name, k)
        elif name.endswith(".npy"):
            assert np.array_equal(np.load(a), np.load(b)), name
        else:
            assert a.read_bytes() == b.read_bytes(), name
