// Action shaping on the device: policy scores -> parent index pairs, one workgroup per env.
//
// `SelectionScores._convert_actions` of the reference (breedgym/vector/vec_wrappers.py:61-79):
//     best  = top_k(scores, k)                                   indices, best first
//     pairs = diallel(best)                                      all (best[i], best[j]), i < j, row-major
//     pick  = choice(key_e, len(pairs), (n_crosses,), replace=False)   = permutation(key_e, P)[:n_crosses]
//     out   = repeat(pairs[pick], ceil(n / n_crosses), total_repeat_length=n)
// done here in one launch: a bitonic sort of (~orderable(score) << 32 | index) for the top-k, jax's
// sort-based shuffle of the P = k(k-1)/2 pair ranks (Threefry words, stable sort = bitonic sort of
// (word << 32 | position)), triangular un-ranking, and the repeat -- nothing leaves LDS in between.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "row_kernels.h"

namespace bh {

// ascending bitonic sort of P (power of two) 64-bit keys in LDS by the whole workgroup
__device__ __forceinline__ void bitonic_sort_lds(uint64_t *keys, int P) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int j = tid; j < (P >> 1); j += nt) {
        const int lo = ((j & ~(stride - 1)) << 1) | (j & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const uint64_t a = keys[lo], b = keys[hi];
        if ((a > b) == up) {
          keys[lo] = b;
          keys[hi] = a;
        }
      }
      __syncthreads();
    }
  }
}

// jax.random.permutation(key, N) into x (LDS, N int32; y is scratch of the same size; keys holds
// P >= N 64-bit words).  Returns the buffer that holds the result (x or y).
__device__ __forceinline__ int32_t *shuffle_lds(u32x2 k, int N, int P, int rounds, uint64_t *keys, int32_t *x,
                                                int32_t *y) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < N; i += nt) x[i] = i;
  const int L = N + (N & 1), h = L >> 1;
  for (int rd = 0; rd < rounds; ++rd) {
    const u32x2 nk = split_at(k.x, k.y, 2u, 0u), sub = split_at(k.x, k.y, 2u, 1u);
    k = nk;
    for (int i = tid; i < h; i += nt) {  // random_bits(sub, N): odd N is padded with a 0 counter
      const bool second = i + h < N;
      const u32x2 o = threefry2x32(sub.x, sub.y, (uint32_t)i, second ? (uint32_t)(i + h) : 0u);
      keys[i] = ((uint64_t)o.x << 32) | (uint32_t)i;
      if (second) keys[i + h] = ((uint64_t)o.y << 32) | (uint32_t)(i + h);
    }
    for (int i = N + tid; i < P; i += nt) keys[i] = ~0ull;
    __syncthreads();
    bitonic_sort_lds(keys, P);
    for (int i = tid; i < N; i += nt) y[i] = x[(uint32_t)keys[i]];
    __syncthreads();
    int32_t *t = x;
    x = y;
    y = t;
  }
  return x;
}

struct SelectionParams {
  const float *scores;  // (E, n)
  int32_t *actions;     // (E, n, 2) out
  uint32_t key0, key1;  // the env key: env e uses split(key, E + 1)[1 + e]
  int32_t E, n, k, n_crosses, n_offspring;
  int32_t P_top;     // power of two >= n
  int32_t n_pairs;   // k (k - 1) / 2
  int32_t P_pairs;   // power of two >= n_pairs
  int32_t rounds;    // shuffle rounds for n_pairs elements
};

__global__ __launch_bounds__(1024) void selection_actions_kernel(const SelectionParams p) {
  extern __shared__ __align__(16) uint64_t sk[];  // max(P_top, P_pairs) keys, 2 x P_pairs int32, k int32
  const int PK = p.P_top > p.P_pairs ? p.P_top : p.P_pairs;
  int32_t *x = reinterpret_cast<int32_t *>(sk + PK), *y = x + p.P_pairs, *best = y + p.P_pairs;
  const int e = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;

  // top-k by score, best first, ties to the lower index (jax.lax.top_k)
  const float *sc = p.scores + (int64_t)e * p.n;
  for (int j = tid; j < p.P_top; j += nt)
    sk[j] = j < p.n ? (((uint64_t)(~orderable(sc[j])) << 32) | (uint32_t)j) : ~0ull;
  __syncthreads();
  bitonic_sort_lds(sk, p.P_top);
  for (int j = tid; j < p.k; j += nt) best[j] = (int32_t)(uint32_t)sk[j];
  __syncthreads();

  // random subset of the diallel pairs, without replacement
  const u32x2 ke = split_at(p.key0, p.key1, (uint32_t)(p.E + 1), (uint32_t)(1 + e));
  const int32_t *perm = shuffle_lds(ke, p.n_pairs, p.P_pairs, p.rounds, sk, x, y);

  // slot s takes pair perm[s / n_offspring]; pair rank t -> (i, j), i < j, row-major upper triangle
  for (int s = tid; s < p.n; s += nt) {
    const int t = perm[s / p.n_offspring];
    // rows before row i hold i (2k - i - 1) / 2 pairs
    int i = (int)((2.0f * p.k - 1.0f - sqrtf((2.0f * p.k - 1.0f) * (2.0f * p.k - 1.0f) - 8.0f * (float)t)) * 0.5f);
    i = i < 0 ? 0 : (i > p.k - 2 ? p.k - 2 : i);
    while (i > 0 && i * (2 * p.k - i - 1) / 2 > t) --i;
    while ((i + 1) * (2 * p.k - i - 2) / 2 <= t) ++i;
    const int j = t - i * (2 * p.k - i - 1) / 2 + i + 1;
    p.actions[((int64_t)e * p.n + s) * 2] = best[i];
    p.actions[((int64_t)e * p.n + s) * 2 + 1] = best[j];
  }
}

}  // namespace bh
