/*
 * breed_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See breed_oracle.h for scope, citations and the "parity unpinned" statement.
 *
 * Written for clarity, not speed: one scalar Threefry call per counter pair,
 * explicit float compare, explicit running XOR.  OpenMP over crosses only.
 */
#include "breed_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* Threefry-2x32-20                                                         */
/* ------------------------------------------------------------------------ */

static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

void bo_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                     uint32_t *o0, uint32_t *o1) {
  static const int R[8] = {13, 15, 26, 6, 17, 29, 16, 24};
  uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
  for (int blk = 0; blk < 5; ++blk) {
    const int *rot = (blk & 1) ? R + 4 : R;
    for (int i = 0; i < 4; ++i) {
      x0 += x1;
      x1 = rotl32(x1, rot[i]);
      x1 ^= x0;
    }
    x0 += ks[(blk + 1) % 3];
    x1 += ks[(blk + 2) % 3] + (uint32_t)(blk + 1);
  }
  *o0 = x0;
  *o1 = x1;
}

void bo_key_from_seed(uint64_t seed, uint32_t key[2]) {
  key[0] = 0u; /* x64 disabled: the high word is shifted out */
  key[1] = (uint32_t)(seed & 0xFFFFFFFFu);
}

/* Which of JAX's two counter layouts split / random_bits follow (jax/_src/prng.py): 0 = the original one
 * (jax_threefry_partitionable = False, default up to jax 0.4.x: _threefry_split_original /
 * _threefry_random_bits_original), 1 = the partitionable one (default since jax 0.5:
 * _threefry_split_foldlike / _threefry_random_bits_partitionable).  Test-infrastructure global. */
static int g_layout = 0;
void bo_set_threefry_layout(int layout) { g_layout = layout ? 1 : 0; }
int bo_threefry_layout(void) { return g_layout; }

/* jax `threefry_2x32(key, count)` for count = iota(n): the counter vector is
 * padded to even length with a 0, cut in two halves (x0 = first, x1 = second),
 * hashed pairwise, and the two output halves are concatenated again. */
static void threefry_iota(const uint32_t key[2], int64_t n, uint32_t *out) {
  int64_t L = n + (n & 1), h = L / 2;
  for (int64_t i = 0; i < h; ++i) {
    uint32_t c0 = (uint32_t)i;
    uint32_t c1 = (i + h < n) ? (uint32_t)(i + h) : 0u; /* the pad counter */
    uint32_t o0, o1;
    bo_threefry2x32(key[0], key[1], c0, c1, &o0, &o1);
    out[i] = o0;
    if (i + h < n) out[i + h] = o1;
  }
}

void bo_random_bits(const uint32_t key[2], int64_t n, uint32_t *out) {
  if (g_layout) { /* element i: counter (hi, lo) = (0, i) of a 64-bit iota, 32-bit draw = o0 ^ o1 */
    for (int64_t i = 0; i < n; ++i) {
      uint32_t o0, o1;
      bo_threefry2x32(key[0], key[1], 0u, (uint32_t)i, &o0, &o1);
      out[i] = o0 ^ o1;
    }
    return;
  }
  threefry_iota(key, n, out);
}

void bo_split(const uint32_t key[2], int64_t num, uint32_t *out) {
  if (g_layout) { /* fold-like: key k = both output words of TF(key, (0, k)) */
    for (int64_t k = 0; k < num; ++k)
      bo_threefry2x32(key[0], key[1], 0u, (uint32_t)k, &out[2 * k], &out[2 * k + 1]);
    return;
  }
  /* counts = iota(2*num); result reshaped (num, 2) */
  threefry_iota(key, 2 * num, out);
}

float bo_uniform_f32(uint32_t bits) {
  uint32_t w = (bits >> 9) | 0x3F800000u;
  float f;
  memcpy(&f, &w, 4);
  f -= 1.0f;
  return f < 0.0f ? 0.0f : f; /* lax.max(minval, ...) */
}

uint32_t bo_threshold23(float r) {
  /* u = k * 2^-23 with k = bits >> 9 (exact in f32).  u < r  <=>  k < r*2^23
   * <=>  k < ceil(r*2^23) for k integer.  r*2^23 is an exact f32 scaling. */
  if (!(r > 0.0f)) return 0u; /* r <= 0 or NaN: never */
  if (r >= 1.0f) return 1u << 23; /* always (k <= 2^23 - 1) */
  return (uint32_t)ceil((double)r * 8388608.0);
}

void bo_meiosis_mask(const uint32_t rec_key[2], const float *r, int64_t M,
                     uint8_t *mask) {
  uint32_t *bits = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(M > 0 ? M : 1));
  bo_random_bits(rec_key, M, bits);
  uint8_t run = 0;
  for (int64_t m = 0; m < M; ++m) {
    uint8_t site = bo_uniform_f32(bits[m]) < r[m];
    run ^= site; /* inclusive cumulative XOR */
    mask[m] = run;
  }
  free(bits);
}

/* ------------------------------------------------------------------------ */
/* permutation / founders / top-k                                           */
/* ------------------------------------------------------------------------ */

typedef struct {
  uint32_t k;
  int32_t v;
  int32_t pos; /* current position: makes qsort stable */
} kv_t;

static int kv_cmp(const void *a, const void *b) {
  const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
  if (x->k != y->k) return x->k < y->k ? -1 : 1;
  return x->pos < y->pos ? -1 : (x->pos > y->pos);
}

void bo_permutation(const uint32_t key_in[2], int64_t N, int32_t *out) {
  /* jax/_src/random.py `_shuffle`: repeated stable sorts by fresh 32-bit keys */
  uint32_t key[2] = {key_in[0], key_in[1]};
  double sz = N > 1 ? (double)N : 1.0;
  int rounds = (int)ceil(3.0 * log(sz) / log(4294967295.0));
  kv_t *kv = (kv_t *)malloc(sizeof(kv_t) * (size_t)(N > 0 ? N : 1));
  uint32_t *bits = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(N > 0 ? N : 1));
  for (int64_t i = 0; i < N; ++i) out[i] = (int32_t)i;
  for (int rd = 0; rd < rounds; ++rd) {
    uint32_t two[4];
    bo_split(key, 2, two);
    key[0] = two[0];
    key[1] = two[1];
    bo_random_bits(&two[2], N, bits);
    for (int64_t i = 0; i < N; ++i) {
      kv[i].k = bits[i];
      kv[i].v = out[i];
      kv[i].pos = (int32_t)i;
    }
    qsort(kv, (size_t)N, sizeof(kv_t), kv_cmp);
    for (int64_t i = 0; i < N; ++i) out[i] = kv[i].v;
  }
  free(kv);
  free(bits);
}

void bo_founders(uint32_t env_key[2], int64_t E, int64_t N, int64_t n, int32_t *out) {
  uint32_t *keys = (uint32_t *)malloc(sizeof(uint32_t) * 2 * (size_t)(E + 1));
  int32_t *perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  bo_split(env_key, E + 1, keys);
  env_key[0] = keys[0];
  env_key[1] = keys[1];
  for (int64_t e = 0; e < E; ++e) {
    bo_permutation(&keys[2 * (e + 1)], N, perm);
    memcpy(out + e * n, perm, sizeof(int32_t) * (size_t)n);
  }
  free(keys);
  free(perm);
}

typedef struct {
  float s;
  int32_t i;
} si_t;

/* jax.lax.top_k orders by XLA's total order on the bit patterns, -NaN < -inf < -0 < +0 < +inf < +NaN (xla's TopK
 * compares the floats as sign-magnitude integers so that the result is deterministic), equal patterns to the lower
 * index.  [EXTERNAL: XLA is not in the reference tree]  Only -0 vs +0 and NaNs differ from a plain float compare. */
static int32_t si_total_order(float f) {
  uint32_t u;
  memcpy(&u, &f, sizeof u);
  return (int32_t)u < 0 ? (int32_t)(0x7FFFFFFFu - u) : (int32_t)u; /* wraps to -1 - |f| for negative patterns */
}

static int si_cmp(const void *a, const void *b) {
  const si_t *x = (const si_t *)a, *y = (const si_t *)b;
  const int32_t vx = si_total_order(x->s), vy = si_total_order(y->s);
  if (vx != vy) return vx > vy ? -1 : 1;
  return x->i < y->i ? -1 : (x->i > y->i);
}

void bo_topk(const float *scores, int64_t n, int64_t k, int32_t *out) {
  si_t *a = (si_t *)malloc(sizeof(si_t) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; ++i) {
    a[i].s = scores[i];
    a[i].i = (int32_t)i;
  }
  qsort(a, (size_t)n, sizeof(si_t), si_cmp);
  for (int64_t i = 0; i < k; ++i) out[i] = a[i].i;
  free(a);
}

/* ------------------------------------------------------------------------ */
/* cross / GEBV / reward                                                    */
/* ------------------------------------------------------------------------ */

int64_t bo_norm_index(int64_t i, int64_t n) {
  /* JAX/NumPy-style retrieval indexing as the reference's gather sees it
   * (vec_env.py:89-90): negative indices count from the end, anything still
   * out of range is clamped (the action space's `high` is inclusive:
   * vec_env.py:63-68, so index n does occur). */
  if (i < 0) i += n;
  if (i < 0) i = 0;
  if (i >= n) i = n - 1;
  return i;
}

int bo_cross(const uint8_t *pool, int64_t E, int64_t n_pool, const int32_t *idx,
             int64_t n_out, int64_t M, const float *r, uint32_t key[2],
             int rng_mode, uint8_t *out, uint8_t *masks) {
  return bo_cross_mutation(pool, E, n_pool, idx, n_out, M, r, key, rng_mode, 0.0f, out, masks);
}

int bo_cross_mutation(const uint8_t *pool, int64_t E, int64_t n_pool, const int32_t *idx,
                      int64_t n_out, int64_t M, const float *r, uint32_t key[2],
                      int rng_mode, float mutation, uint8_t *out, uint8_t *masks) {
  /* [EXTERNAL/UNVERIFIED] chromax.Simulator.cross key plumbing:
   *   keys = split(sim_key, 2*B + 1); sim_key <- keys[0];
   *   gamete (b, p) uses keys[1 + 2*b + p];
   *   inside meiosis: rec_key, mut_key = split(gamete_key, 2);
   *   gamete ^= bernoulli(mut_key, mutation, (M,)) = uniform(mut_key, (M,)) < mutation   (Simulator(mutation=p),
   *   forwarded by the reference through **kwargs: vec_env.py:39,45; 0 by default, then no draw changes anything).
   * B = n_out when the key is shared by all envs (the masks are then computed
   * once per cross slot and applied to every env, as the reference's vmapped
   * graph does), E*n_out otherwise. */
  if (n_pool <= 0 && E * n_out > 0) return -1;

  /* rng_mode bit 1 (value 2) = double haploid [EXTERNAL/UNVERIFIED chromax.Simulator.double_haploid]:
   * keys = split(sim_key, B + 1); row b's single gamete uses keys[1 + b] and fills both slots. */
  int dh = (rng_mode & 2) != 0;
  rng_mode &= 1;
  int64_t B = rng_mode == 0 ? n_out : E * n_out;
  int64_t nkeys = dh ? B + 1 : 2 * B + 1;
  uint32_t *keys = (uint32_t *)malloc(sizeof(uint32_t) * 2 * (size_t)nkeys);
  bo_split(key, nkeys, keys);
  key[0] = keys[0];
  key[1] = keys[1];

  /* pass 1: the 2*B crossover masks */
  uint8_t *mk = (uint8_t *)malloc((size_t)(2 * B * M > 0 ? 2 * B * M : 1));
  uint8_t *mt = mutation > 0.0f ? (uint8_t *)malloc((size_t)(2 * B * M > 0 ? 2 * B * M : 1)) : NULL;
#pragma omp parallel for schedule(static)
  for (int64_t g = 0; g < 2 * B; ++g) {
    uint32_t two[4];
    bo_split(&keys[2 * (1 + (dh ? g / 2 : g))], 2, two);
    bo_meiosis_mask(two, r, M, mk + g * M); /* two[0..1] = rec key */
    if (mt) { /* two[2..3] = mut key */
      uint32_t *bits = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(M > 0 ? M : 1));
      bo_random_bits(&two[2], M, bits);
      for (int64_t m = 0; m < M; ++m) mt[g * M + m] = bo_uniform_f32(bits[m]) < mutation;
      free(bits);
    }
  }

  /* pass 2: gather the parents through idx and pick the alleles the masks name */
  int64_t row = 2 * M;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < E * n_out; ++c) {
    int64_t e = c / n_out, i = c % n_out;
    int64_t b = rng_mode == 0 ? i : c;
    for (int p = 0; p < 2; ++p) {
      const uint8_t *mask = mk + (2 * b + p) * M;
      const uint8_t *par = pool + (e * n_pool + bo_norm_index(idx[2 * c + p], n_pool)) * row;
      uint8_t *o = out + c * row;
      for (int64_t m = 0; m < M; ++m) o[2 * m + p] = par[2 * m + mask[m]];
      if (mt) {
        const uint8_t *flip = mt + (2 * b + p) * M;
        for (int64_t m = 0; m < M; ++m) o[2 * m + p] ^= flip[m];
      }
      if (masks) memcpy(masks + (2 * c + p) * M, mask, (size_t)M);
    }
  }
  free(mk);
  free(mt);
  free(keys);
  return 0;
}

void bo_gebv(const uint8_t *pop, int64_t rows, int64_t M, const float *effects,
             int64_t T, float *out32, double *out64) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < rows; ++i) {
    const uint8_t *g = pop + i * 2 * M;
    for (int64_t t = 0; t < T; ++t) {
      double acc = 0.0;
      for (int64_t m = 0; m < M; ++m) {
        int d = g[2 * m] + g[2 * m + 1];
        acc += (double)d * (double)effects[m * T + t];
      }
      if (out32) out32[i * T + t] = (float)acc;
      if (out64) out64[i * T + t] = acc;
    }
  }
}

void bo_reward_max(const float *gebv, int64_t E, int64_t nT, float *out) {
  for (int64_t e = 0; e < E; ++e) {
    float mx = -INFINITY;
    for (int64_t j = 0; j < nT; ++j)
      if (gebv[e * nT + j] > mx) mx = gebv[e * nT + j];
    out[e] = mx;
  }
}

void bo_optimal_haploid(const uint8_t *pop, int64_t n, int64_t M, const float *effects, int64_t T,
                        const int32_t *block_start, int64_t n_blocks, uint8_t *ohp, float *ohv) {
  uint8_t *h = (uint8_t *)malloc((size_t)(M > 0 ? M : 1));
  for (int64_t i = 0; i < n; ++i) {
    const uint8_t *g = pop + i * M * 2;
    if (n_blocks == 0) {
      for (int64_t m = 0; m < M; ++m) {
        const int a0 = g[2 * m] & 1, a1 = g[2 * m + 1] & 1;
        h[m] = (uint8_t)(effects[m * T] > 0.0f ? (a0 | a1) : (a0 & a1));
      }
    } else {
      for (int64_t b = 0; b < n_blocks; ++b) {
        double s[2] = {0.0, 0.0};
        for (int64_t m = block_start[b]; m < block_start[b + 1]; ++m) {
          double w = 0.0;
          for (int64_t t = 0; t < T; ++t) w += (double)effects[m * T + t];
          s[0] += (g[2 * m] & 1) ? w : 0.0;
          s[1] += (g[2 * m + 1] & 1) ? w : 0.0;
        }
        const int best = s[1] > s[0] ? 1 : 0;
        for (int64_t m = block_start[b]; m < block_start[b + 1]; ++m) h[m] = g[2 * m + best] & 1;
      }
    }
    if (ohp) memcpy(ohp + i * M, h, (size_t)M);
    if (ohv)
      for (int64_t t = 0; t < T; ++t) {
        double acc = 0.0;
        for (int64_t m = 0; m < M; ++m) acc += h[m] ? (double)effects[m * T + t] : 0.0;
        ohv[i * T + t] = (float)(2.0 * acc);
      }
  }
  free(h);
}

void bo_selection_actions(const float *scores, int64_t E, int64_t n, int64_t k,
                          int64_t n_crosses, uint32_t env_key[2], int32_t *out) {
  int64_t P = k * (k - 1) / 2, n_off = (n + n_crosses - 1) / n_crosses;
  uint32_t *keys = (uint32_t *)malloc(sizeof(uint32_t) * 2 * (size_t)(E + 1));
  int32_t *best = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
  int32_t *pairs = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(P > 0 ? P : 1));
  int32_t *perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(P > 0 ? P : 1));
  bo_split(env_key, E + 1, keys);
  env_key[0] = keys[0];
  env_key[1] = keys[1];
  for (int64_t e = 0; e < E; ++e) {
    bo_topk(scores + e * n, n, k, best);
    int64_t t = 0;
    for (int64_t i = 0; i < k; ++i)
      for (int64_t j = i + 1; j < k; ++j, ++t) {
        pairs[2 * t] = best[i];
        pairs[2 * t + 1] = best[j];
      }
    bo_permutation(&keys[2 * (e + 1)], P, perm);
    for (int64_t s = 0; s < n; ++s) {
      int32_t q = perm[s / n_off];
      out[(e * n + s) * 2] = pairs[2 * q];
      out[(e * n + s) * 2 + 1] = pairs[2 * q + 1];
    }
  }
  free(keys);
  free(best);
  free(pairs);
  free(perm);
}

int bo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void bo_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
