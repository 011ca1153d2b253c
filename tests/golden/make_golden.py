"""Generates the committed fixtures under tests/golden/.  Run HERE (build container) only:

    python tests/golden/make_golden.py

Sources of truth:
  * kat_published.json -- vectors published outside this repo: the Random123 Threefry-2x32-20
    known-answer tests (Salmon et al., also asserted by JAX's own random_test.py) and two values
    printed in JAX's public documentation for the *original* threefry layouts
    (`random.split(PRNGKey(0))`, `random.uniform(PRNGKey(0))`).  Typed in by hand, not computed.
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
        },
    }
    (HERE / "kat_published.json").write_text(json.dumps(kat, indent=1))


def one_case(name, pool, idx, r, eff, seed, rng_mode):
    key = co.key_from_seed(seed)
    out, newkey, masks = co.cross(pool, idx, r, key, rng_mode, return_masks=True)
    out_np, newkey_np = no.vec_step_cross(pool, idx, r, key, rng_mode)
    assert np.array_equal(out, out_np) and np.array_equal(newkey, newkey_np), name
    g32, g64 = co.gebv(out, eff, want64=True)
    assert np.allclose(g64, no.gebv(out, eff), rtol=1e-12, atol=1e-9)
    E, n_pool, M, _ = pool.shape
    np.savez_compressed(
        HERE / f"{name}.npz",
        pool_bits=np.packbits(pool.reshape(-1)), pool_shape=np.array(pool.shape),
        idx=idx, r=r, effects=eff, seed=np.array(seed), rng_mode=np.array(rng_mode),
        out_bits=np.packbits(out.reshape(-1)), out_shape=np.array(out.shape),
        masks_bits=np.packbits(masks.reshape(-1)), masks_shape=np.array(masks.shape),
        gebv32=g32, gebv64=g64, new_key=newkey,
    )
    print(name, pool.shape, "->", out.shape, "crossovers/gamete",
          float(np.abs(np.diff(masks.astype(np.int8), axis=-1)).sum(-1).mean()))


def main():
    published()
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
    np.save(HERE / "golden_zero_recomb_chr_lens.npy", gm0.chr_lens)

    # (4) founder draws and top-k
    f, k = co.founders(co.key_from_seed(7), 4, 370, 200)
    f2, k2 = no.founders(no.key_from_seed(7), 4, 370, 200)
    assert np.array_equal(f, f2) and np.array_equal(k, k2)
    perm = co.permutation(co.key_from_seed(5), 70000)  # two shuffle rounds
    assert np.array_equal(perm, no.permutation(no.key_from_seed(5), 70000))
    s = rng.standard_normal(1000).astype(np.float32)
    s[17] = s[400] = s[3]  # ties resolve to the lower index
    np.savez_compressed(HERE / "golden_host_rng.npz", founders=f, founders_key=k, perm70000_head=perm[:64],
                        perm70000_sum=np.array(int((perm.astype(np.int64) * np.arange(70000)).sum())),
                        scores=s, top50=co.topk(s, 50))
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
