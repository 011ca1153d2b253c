"""The reference's five seeded known-answer tests, replayed (tests/golden/reference_kats.json holds their values
and loop parameters as data; the loops below are this repository's own words for
/root/reference/tests/test_vec.py:115-133, :136-154, :186-204, tests/test_env.py:103-119,
tests/test_wrappers.py:73-88).

They were recorded with chromax's bundled genome + genetic map (``chromax.sample_data``), which no machine of this
pipeline has: chromax is not installed, and the in-tree analogue ``breedgym/data/wheat_genome.npy`` is a stripped
blob (``.MISSING_LARGE_BLOBS:3``).  So TODAY every value test below SKIPS with "parity unpinned: blob missing" and
pins nothing.  The day the genome appears -- next to its map ``breedgym/data/wheat_genetic_map.csv`` in a reference
checkout, or through ``BREEDGYM_KAT_GENOME`` / ``BREEDGYM_KAT_GENETIC_MAP`` -- they run against the oracle-backed
engine (CPU) and the HIP engine (``-m gpu``) in both threefry layouts, and a pass in one layout pins the whole chain
key plumbing -> meiosis -> GEBV -> reward to the reference's recorded numbers, with no new code.

What runs today (CPU): the fixture is well formed and every loop kind is executed end to end on stand-in data of
the KATs' shapes, so the harness itself is tested code.
"""
import json
import os
from pathlib import Path

import numpy as np
import pytest

import breedgym_amd
from breedgym_amd.genetic_map import read_genetic_map
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

GOLDEN = Path(__file__).resolve().parent / "golden"
KATS = json.loads((GOLDEN / "reference_kats.json").read_text())["kats"]
REF_DATA = Path("/root/reference/breedgym/data")
LAYOUTS = ("original", "partitionable")
MATCHED = {}  # engine kind -> {KAT name: [layouts whose replay reproduced the recorded reward]}


def _inputs():
    """(genome path, genetic-map path) of the KATs, or None when the genome is not on this machine."""
    genome = os.environ.get("BREEDGYM_KAT_GENOME")
    gmap = os.environ.get("BREEDGYM_KAT_GENETIC_MAP")
    for g, m in ((genome, gmap), (REF_DATA / "wheat_genome.npy", REF_DATA / "wheat_genetic_map.csv"),
                 (GOLDEN / "wheat_genome.npy", GOLDEN / "wheat_genetic_map.csv")):
        if g and m and Path(g).exists() and Path(m).exists():
            return Path(g), Path(m)
    return None


def _engine_kwargs(kind, gm):
    if kind == "hip":
        return {}
    from oracle_engine import OracleEngine

    return {"engine": OracleEngine(gm.recombination_vec, gm.marker_effects)}


def replay(kat, genome, gm, engine_kind, layout, **overrides):
    """One KAT's loop on `genome` (an array or a path) and the GeneticMap `gm` -> the reward(s) of its last step.
    `overrides` replace constructor keywords (the stand-in runs shrink the batch: the CPU oracle's top-k over the
    (n, 7, E, n, 7) block of the pair-score loop takes minutes at 4 x 200)."""
    kw = {k: v for k, v in kat["kwargs"].items() if k not in ("initial_population", "genetic_map", "trait_names")}
    kw.update(overrides)
    try:
        return _replay(kat, genome, gm, engine_kind, layout, kw)
    finally:  # the oracle's counter layout is process-wide state
        from oracle import c_oracle as co

        co.set_threefry_layout("original")


def _replay(kat, genome, gm, engine_kind, layout, kw):
    env = breedgym_amd.make(kat["env_id"], initial_population=genome, genetic_map=gm, threefry_layout=layout,
                            **_engine_kwargs(engine_kind, gm), **kw)
    loop = kat["loop"]
    if loop["kind"] == "random_actions":  # test_vec.py:125-130
        np.random.seed(seed=loop["numpy_seed"])
        pop, _ = env.reset(seed=kat["reset_seed"])
        assert len(pop) == kw["num_envs"]  # = loop["randint_high"] at the KAT's own shape
        for _ in range(loop["steps"]):
            action = np.random.randint(len(pop), size=(kw["num_envs"], kw["individual_per_gen"], 2))
            pop, rews, _, _, _ = env.step(action)
    elif loop["kind"] == "gebv_scores":  # test_vec.py:148-150
        _, infos = env.reset(seed=kat["reset_seed"])
        for _ in range(loop["steps"]):
            _, rews, _, _, infos = env.step(np.asarray(infos["GEBV"]).squeeze())
    elif loop["kind"] == "gebv_outer_sum":  # test_vec.py:196-200
        _, infos = env.reset(seed=kat["reset_seed"])
        for _ in range(loop["steps"]):
            gebvs = np.asarray(infos["GEBV"]).squeeze()
            _, rews, _, _, infos = env.step(np.add.outer(gebvs, gebvs))
    elif loop["kind"] == "fixed_action":  # test_env.py:111-117, test_wrappers.py:82-86
        env.reset(seed=kat["reset_seed"])
        action = loop["action"] if np.isscalar(loop["action"]) else np.array(loop["action"])
        for _ in range(loop["steps"]):
            _, rews, _, _, _ = env.step(action)
    else:
        raise AssertionError(f"unknown loop kind {loop['kind']!r}")
    return np.asarray(rews, dtype=np.float64)


def _map_for(kat, path):
    return read_genetic_map(path, kat["kwargs"].get("trait_names"))


def _check(kat, engine_kind):
    found = _inputs()
    if found is None:
        pytest.skip("parity unpinned: blob missing (chromax.sample_data.genome / breedgym/data/wheat_genome.npy); "
                    "set BREEDGYM_KAT_GENOME and BREEDGYM_KAT_GENETIC_MAP to replay the reference's recorded rewards")
    genome, gmap = found
    got = {}
    for layout in LAYOUTS:
        got[layout] = replay(kat, np.load(genome), _map_for(kat, gmap), engine_kind, layout)
    want = np.asarray(kat["expected"], dtype=np.float64)
    ok = {lay: bool(np.allclose(v, want, rtol=kat["rtol"], atol=kat["atol"])) for lay, v in got.items()}
    # the reference's numbers were recorded under ONE jax version, i.e. one counter layout: either pins the chain --
    # but the SAME one for all five (a mix would be a bug, not a pass): recorded per KAT, checked by the test below
    assert any(ok.values()), f"{kat['name']} ({kat['reference']}): expected {want}, got {got}"
    matched = [lay for lay in LAYOUTS if ok[lay]]
    MATCHED.setdefault(engine_kind, {})[kat["name"]] = matched
    print(f"KAT {kat['name']} ({engine_kind} engine): matched under layout {' and '.join(matched)}")  # shown with -rA


# ---- today: the fixture and the harness ---------------------------------------------------------------------------
def test_fixture_is_well_formed():
    assert [k["name"] for k in KATS] == ["vec_deterministic", "vec_gebv_policy", "vec_pair_score", "env_deterministic",
                                         "kbest_gebv_policy"]
    for k in KATS:
        assert k["env_id"].split(":")[-1] in breedgym_amd.REGISTRY
        assert k["reference"].startswith("tests/") and k["reset_seed"] == 7
        assert np.all(np.isfinite(np.asarray(k["expected"], dtype=np.float64)))
    assert len(KATS[0]["expected"]) == KATS[0]["kwargs"]["num_envs"] == 4


@pytest.mark.parametrize("kat", KATS, ids=[k["name"] for k in KATS])
def test_replay_runs_on_stand_in_data(kat):
    """Every loop end to end on a synthetic 7-trait stand-in (oracle-backed engine; the KAT's own batch shape except
    for the pair-score loop): the harness is executable, deterministic per layout, and the two layouts draw
    different streams."""
    gm7 = synthetic_genetic_map(96, n_chr=3, n_traits=7, seed=11)
    names = kat["kwargs"].get("trait_names")
    if names:  # the wheat map's "Yield" is trait 4 of 7: the stand-in keeps one column
        from breedgym_amd.genetic_map import GeneticMap

        gm = GeneticMap(gm7.recombination_vec, np.ascontiguousarray(gm7.marker_effects[:, 4:5]), list(names), gm7.chr_lens)
    else:
        gm = gm7
    geno = synthetic_genotypes(260, 96, seed=12)
    small = {"num_envs": 2, "individual_per_gen": 24} if kat["loop"]["kind"] == "gebv_outer_sum" else {}
    a = replay(kat, geno, gm, "oracle", "original", **small)
    b = replay(kat, geno, gm, "oracle", "original", **small)
    c = replay(kat, geno, gm, "oracle", "partitionable", **small)
    assert a.shape == (small.get("num_envs"),) if small else a.shape == np.asarray(kat["expected"], dtype=np.float64).shape
    assert np.array_equal(a, b) and np.all(np.isfinite(a)) and not np.array_equal(a, c)


# ---- the day the genome is there ----------------------------------------------------------------------------------
@pytest.mark.parametrize("kat", KATS, ids=[k["name"] for k in KATS])
def test_reference_kat_oracle_engine(kat):
    _check(kat, "oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("kat", KATS, ids=[k["name"] for k in KATS])
def test_reference_kat_hip_engine(kat):
    _check(kat, "hip")


def test_reference_kats_agree_on_one_layout():
    """Runs behind the replays above (file order).  The five rewards were recorded under one jax version: there must be
    ONE layout every replayed KAT matched under, the same for both engines."""
    if not MATCHED:
        pytest.skip("parity unpinned: no KAT was replayed (blob missing)")
    for engine_kind, per_kat in MATCHED.items():
        common = set(LAYOUTS)
        for name, layouts in per_kat.items():
            common &= set(layouts)
        assert common, f"{engine_kind} engine: the KATs do not agree on a threefry layout: {per_kat}"
    print("reference KATs pinned under layout:", {k: sorted(set.intersection(*map(set, v.values()))) for k, v in MATCHED.items()})
