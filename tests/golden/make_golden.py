"""Generates the committed fixtures under tests/golden/.  Run HERE (build container) only:

    python tests/golden/make_golden.py

Sources of truth:
  * kat_published.json -- vectors published outside this repo: the Random123 Threefry-2x32-20
    known-answer tests (Salmon et al., also asserted by JAX's own random_test.py) and two values
    printed in JAX's public documentation for both threefry counter layouts (PRNGKey(0) and the
    tutorial's key(42); the JSON's "_note" names the sources).  Typed in by hand, not computed.
  * reference_kats.json -- the expected values and loop parameters of the reference's five seeded
    known-answer tests, typed in from /root/reference/tests (data; the replay is tests/test_reference_kats.py).
  * golden_*.npz -- inputs plus the outputs of the CPU oracle (C), generated only after the
    NumPy oracle reproduced them bit for bit.  The reference itself cannot produce vectors
    here: chromax / jax / gymnasium are not installed (ModuleNotFoundError), see SURVEY.md 8c.
    `golden_sample` / `golden_zero_recomb` take their INPUT data from the reference's shipped
    sample files (breedgym/data/sample_full_pop_geno.npy, sample_with_r_genetic_map.txt,
    sample_genetic_map.txt): data, not source.
"""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))

from breedgym_amd.genetic_map import read_genetic_map  # noqa: E402
from oracle import c_oracle as co  # noqa: E402
from oracle import numpy_oracle as no  # noqa: E402

REF_DATA = Path("/root/reference/breedgym/data")
OUT = HERE  # main(out_dir) writes elsewhere: tests/test_golden_regen.py regenerates into a scratch directory


def published():
    kat = {
        "threefry2x32_20": [
            {"key": [0, 0], "ctr": [0, 0], "out": [0x6B200159, 0x99BA4EFE]},
            {"key": [0xFFFFFFFF, 0xFFFFFFFF], "ctr": [0xFFFFFFFF, 0xFFFFFFFF], "out": [0x1CB996FC, 0xBB002BE7]},
            {"key": [0x13198A2E, 0x03707344], "ctr": [0x243F6A88, 0x85A308D3], "out": [0xC4923A9C, 0x483DF7A0]},
        ],
        "jax_docs_original_layout": {
            "split_PRNGKey0_num2": [[4146024105, 967050713], [2718843009, 1272950319]],
            "uniform_PRNGKey0": 0.41845703,
            "normal_PRNGKey0": -0.20584226,
            # the "Pseudorandom numbers" tutorial's key(42), as printed under jax <= 0.4.x
            "split_key42_num2": [[2465931498, 3679230171], [255383827, 267815257]],
            "normal_key42": -0.18471177,
            "normal_subkey_of_key42": 1.3694694,
        },
        "_note": "Typed in from published sources, not computed here: Random123's Threefry-2x32-20 known-answer tests; "
                 "values printed in JAX's public documentation (the 'Pseudorandom numbers' tutorial and the jax.random "
                 "module page) -- 'jax_docs_original_layout' under jax <= 0.4.x (jax_threefry_partitionable = False), "
                 "'jax_docs_partitionable_layout' under jax >= 0.5 (the flag's new default).  The first Random123 vector "
                 "is split(key(0))[0] of the partitionable layout by construction.",
        "jax_docs_partitionable_layout": {
            "split_key0_num2": [[1797259609, 2579123966], [928981903, 3453687069]],
            "normal_key42": -0.028304616,
        },
    }
    (OUT / "kat_published.json").write_text(json.dumps(kat, indent=1))


def reference_kats():
    """The reference's five seeded known-answer tests AS DATA (typed in from the reference's test files; values and
    loops only -- tests/test_reference_kats.py holds the replay).  They were recorded with chromax's bundled genome
    and genetic map (`from chromax.sample_data import genetic_map, genome`), which this machine does not have."""
    common = {"initial_population": "<genome>", "genetic_map": "<genetic_map>"}
    kats = {
        "_note": "Expected rewards of the reference's seeded tests, typed in from /root/reference/tests (values and loop "
                 "parameters: data).  Inputs: chromax.sample_data.genome / genetic_map (absent here; the in-tree "
                 "analogue breedgym/data/wheat_genome.npy is a stripped blob, .MISSING_LARGE_BLOBS:3, its map "
                 "breedgym/data/wheat_genetic_map.csv is present: 7 traits, the KATs' (n, 7) GEBV shape).  Until a "
                 "genome file is supplied (BREEDGYM_KAT_GENOME / BREEDGYM_KAT_GENETIC_MAP) nothing here pins anything: "
                 "parity unpinned.",
        "kats": [
            {"name": "vec_deterministic", "reference": "tests/test_vec.py:115-133", "env_id": "VecBreedGym",
             "kwargs": {**common, "num_envs": 4, "individual_per_gen": 200}, "reset_seed": 7,
             "loop": {"kind": "random_actions", "numpy_seed": 7, "steps": 20, "randint_high": 4,
                      "randint_high_is": "len(pop) with pop of shape (num_envs, n, M, 2) = num_envs",
                      "size": [4, 200, 2]},
             "expected": [10.844662, 8.436224, 8.759013, 9.1480465], "rtol": 1e-5, "atol": 1e-8},
            {"name": "vec_gebv_policy", "reference": "tests/test_vec.py:136-154", "env_id": "SelectionScores",
             "kwargs": {**common, "k": 10, "num_envs": 4, "individual_per_gen": 200, "trait_names": ["Yield"]},
             "reset_seed": 7, "loop": {"kind": "gebv_scores", "steps": 10},
             "expected": [18.514063, 19.428415, 19.090204, 19.841211], "rtol": 1e-5, "atol": 1e-8},
            {"name": "vec_pair_score", "reference": "tests/test_vec.py:186-204", "env_id": "PairScores",
             "kwargs": {**common, "num_envs": 4, "individual_per_gen": 200}, "reset_seed": 7,
             "loop": {"kind": "gebv_outer_sum", "steps": 10, "squeeze_note": "infos['GEBV'].squeeze() of (4, 200, T)"},
             "expected": [8.650201, 9.414795, 8.115064, 10.0190525], "rtol": 1e-5, "atol": 1e-8},
            {"name": "env_deterministic", "reference": "tests/test_env.py:103-119", "env_id": "breedgym:BreedGym",
             "kwargs": {**common, "reward_shaping": False}, "reset_seed": 7,
             "loop": {"kind": "fixed_action", "steps": 10,
                      "action": [[1, 2], [1, 5], [1, 7], [2, 5], [2, 9], [4, 7], [4, 8], [5, 9], [6, 8], [6, 9]]},
             "expected": 1.8197979, "rtol": 0.0, "atol": 1e-6},
            {"name": "kbest_gebv_policy", "reference": "tests/test_wrappers.py:73-88", "env_id": "breedgym:KBestBreedGym",
             "kwargs": {**common, "individual_per_gen": 200, "trait_names": ["Yield"]}, "reset_seed": 7,
             "loop": {"kind": "fixed_action", "steps": 10, "action": 10},
             "expected": 18.826467, "rtol": 0.0, "atol": 1e-5},
        ],
    }
    (OUT / "reference_kats.json").write_text(json.dumps(kats, indent=1))


def one_case(name, pool, idx, r, eff, seed, rng_mode):
    key = co.key_from_seed(seed)
    out, newkey, masks = co.cross(pool, idx, r, key, rng_mode, return_masks=True)
    out_np, newkey_np = no.vec_step_cross(pool, idx, r, key, rng_mode)
    assert np.array_equal(out, out_np) and np.array_equal(newkey, newkey_np), name
    g32, g64 = co.gebv(out, eff, want64=True)
    assert np.allclose(g64, no.gebv(out, eff), rtol=1e-12, atol=1e-9)
    E, n_pool, M, _ = pool.shape
    np.savez_compressed(
        OUT / f"{name}.npz",
        pool_bits=np.packbits(pool.reshape(-1)), pool_shape=np.array(pool.shape),
        idx=idx, r=r, effects=eff, seed=np.array(seed), rng_mode=np.array(rng_mode),
        out_bits=np.packbits(out.reshape(-1)), out_shape=np.array(out.shape),
        masks_bits=np.packbits(masks.reshape(-1)), masks_shape=np.array(masks.shape),
        gebv32=g32, gebv64=g64, new_key=newkey,
    )
    print(name, pool.shape, "->", out.shape, "crossovers/gamete",
          float(np.abs(np.diff(masks.astype(np.int8), axis=-1)).sum(-1).mean()))


def main(out_dir=None):
    global OUT
    OUT = Path(out_dir) if out_dir else HERE
    published()
    reference_kats()
    rng = np.random.default_rng(20241003)

    # (1) small: 8 individuals x 256 markers, 3 chromosomes, 2 envs x 8 crosses, 2 traits
    M = 256
    r = (rng.random(M) * 0.05).astype(np.float32)
    r[[0, 100, 180]] = 0.5
    r[5:9] = 0.0
    pool = rng.integers(0, 2, (2, 8, M, 2), dtype=np.uint8)
    idx = rng.integers(0, 8, (2, 8, 2), dtype=np.int32)
    eff = rng.laplace(0, 6, (M, 2)).astype(np.float32)
    one_case("golden_small_shared", pool, idx, r, eff, 7, 0)
    one_case("golden_small_independent", pool, idx, r, eff, 7, 1)

    # (1b) odd marker count (the padded-counter path of random_bits) and M < 64
    M = 61
    r = (rng.random(M) * 0.2).astype(np.float32)
    r[0] = 0.5
    pool = rng.integers(0, 2, (1, 6, M, 2), dtype=np.uint8)
    idx = rng.integers(0, 6, (1, 9, 2), dtype=np.int32)
    eff = rng.standard_normal((M, 1)).astype(np.float32)
    one_case("golden_odd61", pool, idx, r, eff, 123456789, 0)

    # (2) the reference's shipped sample population x its sample map with recombination rates
    geno = np.load(REF_DATA / "sample_full_pop_geno.npy").astype(np.uint8)  # (370, 1000, 2)
    gm = read_genetic_map(REF_DATA / "sample_with_r_genetic_map.txt")
    idx = rng.integers(0, len(geno), (1, 64, 2), dtype=np.int32)
    one_case("golden_sample", geno[None], idx, gm.recombination_vec, gm.marker_effects, 7, 0)

    # (3) all-zero RecombRate map: whole chromosomes are inherited intact
    gm0 = read_genetic_map(REF_DATA / "sample_genetic_map.txt")
    one_case("golden_zero_recomb", geno[None, :64], rng.integers(0, 64, (1, 32, 2), dtype=np.int32),
             gm0.recombination_vec, gm0.marker_effects, 99, 0)
    np.save(OUT / "golden_zero_recomb_chr_lens.npy", gm0.chr_lens)

    # (4) founder draws and top-k
    f, k = co.founders(co.key_from_seed(7), 4, 370, 200)
    f2, k2 = no.founders(no.key_from_seed(7), 4, 370, 200)
    assert np.array_equal(f, f2) and np.array_equal(k, k2)
    perm = co.permutation(co.key_from_seed(5), 70000)  # two shuffle rounds
    assert np.array_equal(perm, no.permutation(no.key_from_seed(5), 70000))
    s = rng.standard_normal(1000).astype(np.float32)
    s[17] = s[400] = s[3]  # ties resolve to the lower index
    np.savez_compressed(OUT / "golden_host_rng.npz", founders=f, founders_key=k, perm70000_head=perm[:64],
                        perm70000_sum=np.array(int((perm.astype(np.int64) * np.arange(70000)).sum())),
                        scores=s, top50=co.topk(s, 50))
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
