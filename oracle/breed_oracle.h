/*
 * breed_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the arithmetic behind the hot path of younik/breedgym
 * (vectorised reset()/step(): parent gather -> meiosis/cross -> GEBV -> reward).
 *
 * PARITY UNPINNED: the reference keeps none of this arithmetic in its own tree.
 * It lives in the un-vendored, un-pinned PyPI packages `chromax` and `jax`
 * (/root/reference/pyproject.toml:24), neither of which is installed in the
 * build container, and the reference's seeded known-answer tests
 * (/root/reference/tests/test_vec.py:115-133,136-154,186-204,
 * tests/test_env.py:103-119) depend on chromax's bundled genome, which is
 * absent as well.  What this file restates is therefore
 *   (1) the reference's own control flow / layouts / formulas, cited per
 *       function below, and
 *   (2) the *published* algorithms of the missing dependencies: JAX's
 *       Threefry-2x32-20 counter PRNG with its "original" (non-partitionable)
 *       split / random_bits / uniform / shuffle layouts, and chromax's
 *       per-marker Bernoulli(r) -> prefix-XOR -> allele-select meiosis.
 * The Threefry primitive is pinned on the Random123 known-answer vectors and
 * the key-split / uniform layouts on values printed in JAX's public docs (see
 * tests/test_oracle_kat.py); everything that needs chromax itself is not
 * pinned and is marked [EXTERNAL/UNVERIFIED] here.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libbreedhip.so) never links or calls it.
 */
#ifndef BREED_ORACLE_H
#define BREED_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- JAX threefry PRNG (original / non-partitionable layouts) ---------- */

/* Threefry-2x32, 20 rounds (Salmon et al., SC'11; jax/_src/prng.py
 * `_threefry2x32_lowering`).  Pure function. */
void bo_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                     uint32_t *o0, uint32_t *o1);

/* jax.random.key(seed) with x64 disabled: (0, seed mod 2^32). */
void bo_key_from_seed(uint64_t seed, uint32_t key[2]);

/* jax.random.split(key, num) -> out[num][2]. */
void bo_split(const uint32_t key[2], int64_t num, uint32_t *out);

/* jax random_bits(key, 32, (n,)) -> out[n] (odd n padded with a 0 counter). */
void bo_set_threefry_layout(int layout); /* 0 original, 1 partitionable (jax >= 0.5 default) */
int bo_threefry_layout(void);
void bo_random_bits(const uint32_t key[2], int64_t n, uint32_t *out);

/* jax.random.uniform float32 in [0,1) from one 32-bit word. */
float bo_uniform_f32(uint32_t bits);

/* integer threshold t such that  uniform(bits) < r  <=>  (bits >> 9) < t. */
uint32_t bo_threshold23(float r);

/* jax.random.permutation(key, N) (sort-based `_shuffle`) -> out[N]. */
void bo_permutation(const uint32_t key[2], int64_t N, int32_t *out);

/* ---- chromax arithmetic [EXTERNAL/UNVERIFIED restatement] -------------- */

/* One gamete's crossover mask: sites = uniform(rec_key, M) < r;
 * mask = inclusive prefix-XOR(sites).  mask[m] in {0,1}. */
void bo_meiosis_mask(const uint32_t rec_key[2], const float *r, int64_t M,
                     uint8_t *mask);

/* Vectorised gather + cross.
 *   pool : (E, n_pool, M, 2) u8 in {0,1}
 *   idx  : (E, n_out, 2) int32, env-local parent rows
 *   out  : (E, n_out, M, 2) u8      out[e,i,m,p] = pool[e, idx[e,i,p], m, mask_p[m]]
 *   masks: optional (E, n_out, 2, M) u8 crossover masks (may be NULL)
 *   key  : simulator key, advanced in place exactly once per call
 *   rng_mode 0: every env sees the same keys (what jax.vmap over an
 *               un-batched simulator key does: vec_env.py:75-77)
 *            1: the E*n_out crosses are one flat chromax cross() batch
 * Parent indices follow JAX retrieval indexing: negative wraps once, then clamp.
 * Returns 0, or -1 if a pool is empty. */
int64_t bo_norm_index(int64_t i, int64_t n);
int bo_cross(const uint8_t *pool, int64_t E, int64_t n_pool, const int32_t *idx,
             int64_t n_out, int64_t M, const float *r, uint32_t key[2],
             int rng_mode, uint8_t *out, uint8_t *masks);
/* The same with chromax's mutation probability [EXTERNAL/UNVERIFIED]: every gamete allele is flipped where
 * uniform(mut_key, (M,)) < mutation, mut_key = split(gamete_key, 2)[1].  `masks` stay the crossover masks. */
int bo_cross_mutation(const uint8_t *pool, int64_t E, int64_t n_pool, const int32_t *idx,
                      int64_t n_out, int64_t M, const float *r, uint32_t key[2],
                      int rng_mode, float mutation, uint8_t *out, uint8_t *masks);

/* GEBV[i,t] = sum_m (pop[i,m,0] + pop[i,m,1]) * effects[m,t]
 * (scripts/optimize_diff_cross.py:16-17, utils/index_functions.py:55).
 * Accumulated in f64 in marker order; out32 = (float)sum, out64 optional. */
void bo_gebv(const uint8_t *pop, int64_t rows, int64_t M, const float *effects,
             int64_t T, float *out32, double *out64);

/* reward = max over (individuals, traits) per env (vec_env.py:96-98). */
void bo_reward_max(const float *gebv, int64_t E, int64_t nT, float *out);

/* VecBreedGym.reset founder draw (vec_env.py:22-27,120-128): keys =
 * split(env_key, E+1); env_key <- keys[0]; env e draws
 * permutation(keys[1+e], N)[:n].  out (E, n) int32. */
void bo_founders(uint32_t env_key[2], int64_t E, int64_t N, int64_t n,
                 int32_t *out);

/* jax.lax.top_k indices: k largest, descending, ties -> lower index first. */
void bo_topk(const float *scores, int64_t n, int64_t k, int32_t *out);

/* SelectionScores._convert_actions for E envs (vec_wrappers.py:61-85): keys =
 * split(env_key, E+1); env_key <- keys[0]; env e: best = topk(scores[e], k);
 * pairs = all (best[i], best[j]), i < j, row-major; pick =
 * permutation(keys[1+e], len(pairs))[:n_crosses]; out = repeat(pairs[pick],
 * ceil(n / n_crosses), total_repeat_length = n).  out (E, n, 2) int32. */
void bo_selection_actions(const float *scores, int64_t E, int64_t n, int64_t k,
                          int64_t n_crosses, uint32_t env_key[2], int32_t *out);

/* Optimal haploid of every individual (utils/index_functions.py:36-78) and its value (:21-33).
 *   pop (n, M, 2) u8; effects (M, T) f32; block_start (n_blocks + 1) marker offsets, or
 *   n_blocks = 0 for the blockless rule: OHP[m] = any(alleles) where effects[m, 0] > 0, else
 *   all(alleles) (:66-78; `positive_mask` is chromax's, [EXTERNAL/UNVERIFIED]: effect > 0).
 *   With blocks (:44-63): per block the haplotype d with the larger sum_m sum_t pop[i,m,d] *
 *   effects[m,t] (first on ties, like argmax) supplies the block.
 *   ohp (n, M) u8 or NULL; ohv (n, T) f32 or NULL = 2 * sum_m OHP[i,m] * effects[m,t], f64 sums. */
void bo_optimal_haploid(const uint8_t *pop, int64_t n, int64_t M, const float *effects, int64_t T,
                        const int32_t *block_start, int64_t n_blocks, uint8_t *ohp, float *ohv);

/* number of OpenMP threads the batch functions will use */
int bo_num_threads(void);
void bo_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
