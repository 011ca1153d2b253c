"""Consumes ``tests/golden/chromax_pinned.npz`` -- what chromax / JAX / the reference returned for ten groups of
inputs, written by ``tools/pin_with_chromax.py`` on a machine that HAS chromax -- and replays the same inputs through
``breedgym_amd``: the oracle-backed engine on the CPU, the HIP engine under ``-m gpu``.  One maintainer run pins every
[EXTERNAL/UNVERIFIED] item of DESIGN.md section 3 at once.

The file cannot be made in this pipeline (no chromax, no jax, no network).  While it is absent the value tests SKIP with
"parity unpinned" and pin nothing; what runs regardless: the script's inputs are reproducible from numpy alone, and the
replay below executes end to end on them (so the harness is tested code, as tests/test_reference_kats.py's is).
"""
import importlib.util
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

ROOT = Path(__file__).resolve().parents[1]
PINNED = ROOT / "tests" / "golden" / "chromax_pinned.npz"
UNPINNED = ("parity unpinned: tests/golden/chromax_pinned.npz is not there -- tools/pin_with_chromax.py writes it on a machine "
            "with chromax installed (not this one)")


def _script():
    spec = importlib.util.spec_from_file_location("pin_with_chromax", ROOT / "tools" / "pin_with_chromax.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _gmap(d):
    g = pd.DataFrame({"CHR.PHYS": d["chrom"], "RecombRate": d["recomb"]})
    for t, name in enumerate(("Yield", "Height", "Protein")):
        g[name] = d["effects"][:, t]
    return g


def replay(d, seeds, mutation_p, layout, engine_kind, env_actions):
    """The ten items of tools/pin_with_chromax.py on inputs `d`, through breedgym_amd -> {name: array}."""
    import torch

    import breedgym_amd
    from breedgym_amd import Simulator
    from breedgym_amd.genetic_map import read_genetic_map
    from breedgym_amd.vector import PairScores, SelectionScores, VecBreedGym

    gm = read_genetic_map(_gmap(d))

    def kw():
        if engine_kind == "hip":
            return {"threefry_layout": layout}
        from oracle_engine import OracleEngine

        return {"threefry_layout": layout, "engine": OracleEngine(gm.recombination_vec, gm.marker_effects)}

    import warnings

    s_cross, s_mut, s_dh, s_env = (int(x) for x in seeds)
    geno = d["geno"].astype(bool)
    parents = geno[d["parents_idx"]]
    out = {}
    sim = Simulator(genetic_map=gm, seed=s_cross, **kw())
    out["held_recombination_vec"] = np.asarray(sim.recombination_vec, np.float32)
    out["cross_out"] = np.asarray(sim.cross(parents)).astype(np.uint8)
    out["cross_key_after"] = np.asarray(sim.random_key, np.uint32)
    out["cross_out_2"] = np.asarray(sim.cross(parents)).astype(np.uint8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        simm = Simulator(genetic_map=gm, seed=s_mut, mutation=float(mutation_p), **kw())
    out["mutation_out"] = np.asarray(simm.cross(parents)).astype(np.uint8)
    simd = Simulator(genetic_map=gm, seed=s_dh, **kw())
    out["dh_out"] = np.asarray(simd.double_haploid(simd.device_put(geno[:10]), n_offspring=3)).astype(np.uint8)
    out["dh_key_after"] = np.asarray(simd.random_key, np.uint32)
    pop = sim.device_put(geno)
    sp, si = sim.select(pop, k=7)
    out["select_idx"], out["select_pop"] = np.asarray(si, np.int32), np.asarray(sp).astype(np.uint8)
    n = geno.shape[0]
    bp, bi = sim.select(sim.device_put(geno.reshape(4, n // 4, *geno.shape[1:])), k=5)
    out["select_batched_idx"], out["select_batched_pop"] = np.asarray(bi, np.int32), np.asarray(bp).astype(np.uint8)
    out["diallel"] = np.asarray(sim._diallel_indices(np.array([5, 2, 9, 0, 7])), np.int32)
    out["gebv_model"] = np.asarray(sim.GEBV_model(pop), np.float32)
    frame = sim.GEBV(pop)
    out["gebv_frame"], out["gebv_frame_columns"] = np.asarray(frame, np.float32), np.array(list(frame.columns))
    out["held_marker_effects"] = np.asarray(sim.GEBV_model.marker_effects, np.float32)
    out["positive_mask"] = np.asarray(sim.GEBV_model.positive_mask).astype(np.uint8)

    n_pair = d["pair_scores"].shape[-1]
    env_p = VecBreedGym(num_envs=d["pair_scores"].shape[0], initial_population=geno, genetic_map=gm, individual_per_gen=n_pair, **kw())
    out["pair_actions"] = np.asarray(torch.as_tensor(PairScores(env_p)._convert_actions(d["pair_scores"])).cpu(), np.int32)
    k_sel, nc_sel = 6, 12
    env_s = VecBreedGym(num_envs=4, initial_population=geno, genetic_map=gm, individual_per_gen=n, **kw())
    w = SelectionScores(env_s, k=k_sel, n_crosses=nc_sel)
    env_s.random_key = breedgym_amd.rng.key(s_env)
    out["selection_params"] = np.array([k_sel, nc_sel])
    out["selection_actions"] = np.asarray(torch.as_tensor(w._convert_actions(d["sel_scores"])).cpu(), np.int32)
    env = VecBreedGym(num_envs=4, initial_population=geno, genetic_map=gm, individual_per_gen=12, num_generations=3,
                      reward_shaping=True, autoreset=False, **kw())
    pop0, _ = env.reset(seed=s_env)
    out["env_founders_pop"] = np.asarray(pop0).astype(np.uint8)
    rews, gebvs = [], []
    for a in env_actions:
        p, rew, ter, tru, info = env.step(a)
        rews.append(np.asarray(rew, np.float32))
        gebvs.append(np.asarray(info["GEBV"], np.float32))
    out["env_rewards"], out["env_gebv"], out["env_last_pop"] = np.stack(rews), np.stack(gebvs), np.asarray(p).astype(np.uint8)
    return out


EXACT = ("cross_out", "cross_key_after", "cross_out_2", "mutation_out", "dh_out", "dh_key_after", "select_idx", "select_pop",
         "select_batched_idx", "select_batched_pop", "diallel", "positive_mask", "pair_actions", "selection_actions",
         "env_founders_pop", "env_last_pop", "held_recombination_vec", "held_marker_effects")
CLOSE = ("gebv_model", "gebv_frame", "env_rewards", "env_gebv")  # 1e-5 relative: BASELINE's tolerance


def compare(ours, theirs):
    bad = []
    for k in EXACT:
        if k in theirs and theirs[k].size and not np.array_equal(ours[k], theirs[k]):
            bad.append(f"{k}: {int(np.sum(np.asarray(ours[k]) != np.asarray(theirs[k])))} of {theirs[k].size} differ")
    for k in CLOSE:
        if k in theirs and not np.allclose(ours[k], theirs[k], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(theirs[k]).max()))):
            bad.append(f"{k}: max |d| = {float(np.abs(ours[k] - theirs[k]).max())}")
    if "gebv_frame_columns" in theirs and list(ours["gebv_frame_columns"]) != list(theirs["gebv_frame_columns"]):
        bad.append("gebv_frame_columns")
    return bad


def test_the_pinning_inputs_are_reproducible_and_the_replay_runs():
    """No chromax needed: the script's inputs come from numpy alone (twice the same), and every item replays end to end
    through the oracle-backed engine -- shapes as chromax would return them, doubled haploids doubled, keys advanced."""
    m = _script()
    a, b = m.make_inputs(), m.make_inputs()
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert a["geno"].shape == (m.N, m.M, 2) and a["effects"].shape == (m.M, m.T) and abs(float(a["effects"][:, 0].mean())) > 1
    acts = np.random.default_rng(5).integers(0, 12, (3, 4, 12, 2), dtype=np.int32)
    ours = replay(a, (m.SEED_CROSS, m.SEED_MUT, m.SEED_DH, m.SEED_ENV), m.MUTATION_P, "original", "oracle", acts)
    assert ours["cross_out"].shape == (24, m.M, 2) and not np.array_equal(ours["cross_out"], ours["cross_out_2"])
    assert not np.array_equal(ours["cross_out"], ours["mutation_out"]) or m.MUTATION_P == 0
    assert ours["dh_out"].shape == (10, 3, m.M, 2) and np.array_equal(ours["dh_out"][..., 0], ours["dh_out"][..., 1])
    assert ours["select_idx"].shape == (7,) and ours["select_batched_idx"].shape == (4, 5) and ours["diallel"].shape == (10, 2)
    assert ours["pair_actions"].shape == (4, 16, 2) and ours["selection_actions"].shape == (4, m.N, 2)
    assert ours["env_rewards"].shape == (3, 4) and ours["env_founders_pop"].shape == (4, 12, m.M, 2)
    assert compare(ours, ours) == []


def _pinned():
    if not PINNED.exists():
        pytest.skip(UNPINNED)
    z = np.load(PINNED, allow_pickle=False)
    d = {k[3:]: z[k] for k in z.files if k.startswith("in_") and k != "in_env_actions"}
    return z, d


def test_chromax_pinned_values_oracle_engine():
    z, d = _pinned()
    ours = replay(d, z["seeds"], float(z["mutation_p"]), str(z["layout"]), "oracle", z["in_env_actions"])
    bad = compare(ours, {k: z[k] for k in z.files})
    assert not bad, f"breedgym_amd (CPU oracle engine) differs from chromax ({z['versions']}, layout {z['layout']}): {bad}"


@pytest.mark.gpu
def test_chromax_pinned_values_hip_engine():
    z, d = _pinned()
    ours = replay(d, z["seeds"], float(z["mutation_p"]), str(z["layout"]), "hip", z["in_env_actions"])
    bad = compare(ours, {k: z[k] for k in z.files})
    assert not bad, f"breedgym_amd (HIP engine) differs from chromax ({z['versions']}, layout {z['layout']}): {bad}"
