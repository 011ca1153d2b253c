#!/usr/bin/env python
"""Pin every [EXTERNAL/UNVERIFIED] item of DESIGN.md section 3 in ONE run -- for a maintainer who HAS chromax.

    pip install chromax "jax[cpu]" gymnasium            # the reference's own dependencies (pyproject.toml:24)
    python tools/pin_with_chromax.py --reference /path/to/breedgym [--layout original|partitionable]
    python -m pytest tests/test_chromax_pinned.py -q       # + -m gpu on an MI355X: the HIP engine

This cannot run in the build container (no chromax, no jax, no network): nothing under tests/golden/ was made by it yet,
and until it is, tests/test_chromax_pinned.py skips with "parity unpinned".  It writes
``tests/golden/chromax_pinned.npz``: for each item below the INPUTS (made here with numpy from fixed seeds -- no data
of chromax's is needed) and what chromax / JAX / the reference RETURNED for them, the threefry layout the run was made
under, and the versions.  The consuming test replays the same inputs through ``breedgym_amd`` (oracle-backed engine on
the CPU, the HIP engine under ``-m gpu``) and compares.  One maintainer run pins all of:

  1 cross            Simulator.cross(parents): the 2B+1 key split, gamete key 1+2b+p, split(gamete_key)[0] for the
                     uniforms, sites = uniform < r, inclusive cumulative XOR, r = 0.5 at chromosome starts
  2 mutation         Simulator(mutation=p).cross: which half of the gamete key draws the flips, one uniform per allele
  3 double_haploid   Simulator.double_haploid(pop, n_offspring): the B+1 split, one gamete on both strands
  4 select           Simulator.select(pop, k) and its batched (vmap) form: default index, order, ties
  5 diallel          Simulator._diallel_indices: the order of the pairs
  6 gebv             GEBV_model(pop) on a map whose effects have a non-zero mean ("no offset term"), and .GEBV's frame
  7 positive_mask    GEBV_model.positive_mask (= effect > 0 ?)
  8 pair_scores      the reference's PairScores._convert_actions on scores whose softmax * n sits next to integers
                     (jax.nn.softmax reduces in f32: breedgym_amd sums in f64 -- DESIGN.md section 3)
  9 selection_scores the reference's SelectionScores._convert_actions (top-k, choice without replacement, repeat)
 10 env              VecBreedGym: reset(seed) founders and three steps' rewards -- the whole chain, on these inputs

Items 8-10 import the REFERENCE (``--reference``: a checkout of younik/breedgym, or an installed ``breedgym``).
If a comparison fails afterwards, INTEGRATION.md section 4 says which one function on each side to correct.
"""
import argparse
import sys
import types
from pathlib import Path

import numpy as np
import pandas as pd

ROOT = Path(__file__).resolve().parents[1]
OUT = ROOT / "tests" / "golden" / "chromax_pinned.npz"

M, N_CHR, N, T = 1024, 4, 48, 3          # markers, chromosomes, individuals, traits: small enough for the CPU oracle
SEED_CROSS, SEED_MUT, SEED_DH, SEED_ENV = 11, 12, 13, 14
MUTATION_P = 0.01


def make_inputs():
    """Everything the run feeds to chromax, from numpy alone (seeds fixed; stored in the .npz beside the outputs)."""
    rng = np.random.default_rng(20260105)
    chrom = np.repeat(np.arange(1, N_CHR + 1), M // N_CHR).astype(np.int32)
    r = rng.lognormal(-5.5, 1.0, M)
    r[rng.random(M) < 0.16] = 0.0
    r = np.minimum(r, 0.4)
    # non-zero-mean effects (item 6), heavy tails, a few exact zeros (item 7: is 0 "positive"?)
    eff = rng.laplace(0.0, 6.0, (M, T)).astype(np.float32) + np.array([3.5, -1.25, 0.0], np.float32)
    eff[rng.random((M, T)) < 0.02] = 0.0
    gmap = pd.DataFrame({"CHR.PHYS": chrom, "RecombRate": r.astype(np.float64)})
    for t, name in enumerate(("Yield", "Height", "Protein")):
        gmap[name] = eff[:, t]
    freq = rng.beta(0.5, 2.0, M)
    a0 = rng.random((N, M)) < freq
    a1 = np.where(rng.random((N, M)) < 0.9, a0, rng.random((N, M)) < freq)
    geno = np.stack((a0, a1), axis=-1)                                   # (N, M, 2) bool
    parents_idx = rng.integers(0, N, (24, 2), dtype=np.int32)           # item 1/2: 24 crosses
    # item 8: scores whose softmax * n lies within an ulp or two of an integer: equal scores (sm = 1/n exactly or not),
    # two-level scores, and random ones
    n_pair = 16
    pair = np.stack([
        np.zeros((n_pair, n_pair), np.float32),
        np.where(np.arange(n_pair * n_pair).reshape(n_pair, n_pair) % 4 == 0, np.log(3.0), 0.0).astype(np.float32),
        rng.normal(0, 1, (n_pair, n_pair)).astype(np.float32),
        rng.normal(0, 8, (n_pair, n_pair)).astype(np.float32),
    ])
    sel_scores = rng.normal(0, 1, (4, N)).astype(np.float32)             # item 9 / 10: 4 envs
    sel_scores[1, :8] = sel_scores[1, 8:16]                               # ties
    return dict(chrom=chrom, recomb=r, effects=eff, geno=geno, parents_idx=parents_idx, pair_scores=pair,
                sel_scores=sel_scores)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", help="checkout of younik/breedgym (its root: the directory that holds breedgym/)")
    ap.add_argument("--layout", choices=("original", "partitionable"), default="original",
                    help="jax_threefry_partitionable for this run (recorded; the reference's seeded tests were recorded under 'original')")
    args = ap.parse_args()
    try:
        import jax
        import jax.numpy as jnp
        import chromax
        from chromax import Simulator
    except ImportError as e:
        sys.exit(f"pin_with_chromax.py needs chromax and jax ({e}); it cannot run in the build container -- see the docstring")
    jax.config.update("jax_threefry_partitionable", args.layout == "partitionable")
    if args.reference:
        sys.path.insert(0, str(Path(args.reference).resolve()))

    d = make_inputs()
    gmap = pd.DataFrame({"CHR.PHYS": d["chrom"], "RecombRate": d["recomb"]})
    for t, name in enumerate(("Yield", "Height", "Protein")):
        gmap[name] = d["effects"][:, t]
    out = {f"in_{k}": v for k, v in d.items()}
    out["layout"] = np.array(args.layout)
    out["versions"] = np.array(f"chromax {getattr(chromax, '__version__', '?')}, jax {jax.__version__}, numpy {np.__version__}")
    out["seeds"] = np.array([SEED_CROSS, SEED_MUT, SEED_DH, SEED_ENV])
    out["mutation_p"] = np.float64(MUTATION_P)
    geno = d["geno"]
    parents = geno[d["parents_idx"]]                                     # (24, 2, M, 2), as vec_env.py:89-90 builds them

    def key_words(k):
        return np.asarray(jax.random.key_data(k) if hasattr(jax.random, "key_data") else k, np.uint32)

    # 1 cross
    sim = Simulator(genetic_map=gmap, seed=SEED_CROSS)
    out["held_recombination_vec"] = np.asarray(getattr(sim, "recombination_vec", np.zeros(0)), np.float32)
    out["cross_out"] = np.asarray(sim.cross(jnp.asarray(parents))).astype(np.uint8)
    out["cross_key_after"] = key_words(sim.random_key)
    out["cross_out_2"] = np.asarray(sim.cross(jnp.asarray(parents))).astype(np.uint8)   # second call: the key chain
    # 2 mutation
    simm = Simulator(genetic_map=gmap, seed=SEED_MUT, mutation=MUTATION_P)
    out["mutation_out"] = np.asarray(simm.cross(jnp.asarray(parents))).astype(np.uint8)
    # 3 double haploid
    simd = Simulator(genetic_map=gmap, seed=SEED_DH)
    out["dh_out"] = np.asarray(simd.double_haploid(jnp.asarray(geno[:10]), n_offspring=3)).astype(np.uint8)
    out["dh_key_after"] = key_words(simd.random_key)
    # 4 select (+ batched)
    sel_pop, sel_idx = sim.select(jnp.asarray(geno), k=7)
    out["select_idx"], out["select_pop"] = np.asarray(sel_idx, np.int32), np.asarray(sel_pop).astype(np.uint8)
    bpop, bidx = jax.vmap(lambda p: sim.select(p, k=5))(jnp.asarray(geno.reshape(4, N // 4, M, 2)))
    out["select_batched_idx"], out["select_batched_pop"] = np.asarray(bidx, np.int32), np.asarray(bpop).astype(np.uint8)
    # 5 diallel
    out["diallel"] = np.asarray(sim._diallel_indices(jnp.asarray([5, 2, 9, 0, 7])), np.int32)
    # 6 GEBV
    out["gebv_model"] = np.asarray(sim.GEBV_model(jnp.asarray(geno)), np.float32)
    frame = sim.GEBV(jnp.asarray(geno))
    out["gebv_frame"], out["gebv_frame_columns"] = np.asarray(frame, np.float32), np.array(list(frame.columns))
    out["held_marker_effects"] = np.asarray(getattr(sim.GEBV_model, "marker_effects", np.zeros(0)), np.float32)
    # 7 positive mask
    out["positive_mask"] = np.asarray(getattr(sim.GEBV_model, "positive_mask", np.zeros(0))).astype(np.uint8)

    # 8-10: the reference itself
    try:
        from breedgym.vector import PairScores, SelectionScores, VecBreedGym
    except ImportError as e:
        sys.exit(f"items 8-10 need the reference importable (--reference PATH, gymnasium installed): {e}")
    n_pair = d["pair_scores"].shape[-1]
    stub = types.SimpleNamespace(n_crosses=n_pair)
    out["pair_actions"] = np.asarray(PairScores._convert_actions(stub, jnp.asarray(d["pair_scores"])), np.int32)
    k_sel, nc_sel = 6, 12
    stub = types.SimpleNamespace(k=k_sel, n_crosses=nc_sel, individual_per_gen=N, simulator=sim)
    keys = jax.random.split(jax.random.PRNGKey(SEED_ENV), 4 + 1)
    out["selection_params"] = np.array([k_sel, nc_sel])
    out["selection_actions"] = np.asarray(SelectionScores._convert_actions(stub, jnp.asarray(d["sel_scores"]), keys[1:]), np.int32)
    env = VecBreedGym(num_envs=4, initial_population=geno, genetic_map=gmap, individual_per_gen=12, num_generations=3,
                      reward_shaping=True, autoreset=False)
    pop0, _ = env.reset(seed=SEED_ENV)
    out["env_founders_pop"] = np.asarray(pop0).astype(np.uint8)
    arng = np.random.default_rng(5)
    acts = arng.integers(0, 12, (3, 4, 12, 2), dtype=np.int32)
    out["in_env_actions"] = acts
    rews, gebvs = [], []
    for a in acts:
        pop, rew, ter, tru, info = env.step(jnp.asarray(a))
        rews.append(np.asarray(rew, np.float32))
        gebvs.append(np.asarray(info["GEBV"], np.float32))
    out["env_rewards"], out["env_gebv"], out["env_last_pop"] = np.stack(rews), np.stack(gebvs), np.asarray(pop).astype(np.uint8)

    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes): layout {args.layout}, {out['versions']}")
    print("now: python -m pytest tests/test_chromax_pinned.py -q      (and -m gpu on an MI355X)")


if __name__ == "__main__":
    main()
