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

#include "founders.h"
#include "threefry.h"

namespace bh {

__device__ __forceinline__ uint32_t orderable(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// ---- top-k by bitonic sort in LDS --------------------------------------------------------
// key = (~orderable(score)) << 32 | index : ascending key order = descending score, ties to
// the lower index (jax.lax.top_k).  One block sorts one chunk of up to P (power of two)
// candidates of one row and emits its first kk (index, score) pairs.
struct TopkParams {
  const float *scores;    // (rows, n_in) candidate scores
  const int32_t *in_idx;  // (rows, n_in) original indices, or null (= position)
  int32_t *out_idx;       // (rows, chunks*kk)
  float *out_scores;      // (rows, chunks*kk) or null
  int32_t n_in, chunk, P, kk;
};

__global__ __launch_bounds__(1024) void topk_chunk_kernel(const TopkParams p) {
  extern __shared__ __align__(16) uint64_t keys[];
  const int row = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int base = ch * p.chunk;
  const int cnt = min(p.chunk, p.n_in - base);
  const float *sc = p.scores + (int64_t)row * p.n_in + base;
  const int32_t *ii = p.in_idx ? p.in_idx + (int64_t)row * p.n_in + base : nullptr;
  for (int j = tid; j < p.P; j += nt) {
    uint64_t k = ~0ull;
    if (j < cnt) {
      const uint32_t id = ii ? (uint32_t)ii[j] : (uint32_t)(base + j);
      k = ((uint64_t)(~orderable(sc[j])) << 32) | id;
    }
    keys[j] = k;
  }
  __syncthreads();
  bitonic_sort_lds(keys, p.P);
  const int n_chunks = gridDim.x;
  for (int j = tid; j < p.kk; j += nt) {
    const uint64_t k = keys[j];
    const int64_t o = (int64_t)row * n_chunks * p.kk + (int64_t)ch * p.kk + j;
    const bool valid = j < cnt;
    p.out_idx[o] = valid ? (int32_t)(uint32_t)k : -1;
    if (p.out_scores) {
      // undo the key: orderable -> float
      uint32_t u = ~(uint32_t)(k >> 32);
      u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
      p.out_scores[o] = valid ? __uint_as_float(u) : -INFINITY;
    }
  }
}

__global__ __launch_bounds__(1024) void founders_kernel(const FoundersParams p) {
  extern __shared__ __align__(16) uint64_t fk_lds[];
  founders_role(p, (int)blockIdx.x, fk_lds);
}

// jax.random.permutation(key, N) into x (LDS, N int32; y is scratch of the same size; keys holds
// P >= N 64-bit words).  Returns the buffer that holds the result (x or y).
__device__ __forceinline__ int32_t *shuffle_lds(u32x2 k, int N, int P, int rounds, uint64_t *keys, int32_t *x,
                                                int32_t *y, int layout) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < N; i += nt) x[i] = i;
  // (zero rounds -- N = 1: SelectionScores with k = 2 has ONE pair -- return x as it stands: without this barrier the
  // other waves read x[0] before wave 0 has written it; profiles/r05_fuzz_env_mismatch.txt)
  if (rounds == 0) __syncthreads();
  const int L = N + (N & 1), h = L >> 1;
  for (int rd = 0; rd < rounds; ++rd) {
    const u32x2 nk = split_at(k.x, k.y, 2u, 0u, layout), sub = split_at(k.x, k.y, 2u, 1u, layout);
    k = nk;
    if (layout == LAYOUT_PARTITIONABLE) {
      for (int i = tid; i < N; i += nt)
        keys[i] = ((uint64_t)bits_word_partitionable(sub.x, sub.y, (uint32_t)i) << 32) | (uint32_t)i;
    } else {
      for (int i = tid; i < h; i += nt) {  // random_bits(sub, N): odd N is padded with a 0 counter
        const bool second = i + h < N;
        const u32x2 o = threefry2x32(sub.x, sub.y, (uint32_t)i, second ? (uint32_t)(i + h) : 0u);
        keys[i] = ((uint64_t)o.x << 32) | (uint32_t)i;
        if (second) keys[i + h] = ((uint64_t)o.y << 32) | (uint32_t)(i + h);
      }
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

// ---- top-k of long score vectors: radix select + sort of the winners ---------------------------
// One workgroup per row.  (1) the k-th largest key T by four 8-bit histogram passes, most
// significant digit first; (2) everything above T, plus the lowest-index `need` elements equal to
// T (block scan in index order), is compacted into LDS as (~key << 32 | index); (3) a bitonic sort of
// those k <= 16384 words gives jax.lax.top_k's order: descending score, ties to the lower index.
struct TopkSelectParams {
  const float *scores;  // (rows, n)
  int32_t *out_idx;     // (rows, k)
  int32_t n, k, P;      // P = power of two >= k
};

__global__ __launch_bounds__(1024) void topk_select_kernel(const TopkSelectParams p) {
  extern __shared__ __align__(16) uint64_t sel[];  // P words
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_need, s_count, s_wave[16], s_running;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6;
  const float *sc = p.scores + (int64_t)blockIdx.x * p.n;

  if (tid == 0) {
    s_prefix = 0;
    s_need = (unsigned)p.k;
  }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    for (int j = tid; j < 256; j += nt) hist[j] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    for (int j = tid; j < p.n; j += nt) {
      const unsigned key = orderable(sc[j]);
      if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned need = s_need;
      int d = 255;
      for (; d > 0; --d) {
        if (hist[d] >= need) break;
        need -= hist[d];
      }
      s_need = need;
      s_prefix = (prefix << 8) | (unsigned)d;
    }
    __syncthreads();
  }
  const unsigned T = s_prefix, need = s_need;  // `need` of the elements equal to T are taken
  const unsigned n_gt = (unsigned)p.k - need;

  if (tid == 0) {
    s_count = 0;
    s_running = 0;
  }
  for (int j = tid; j < p.P; j += nt) sel[j] = ~0ull;
  __syncthreads();
  // above the threshold: any order (they are sorted afterwards)
  for (int j = tid; j < p.n; j += nt) {
    const unsigned key = orderable(sc[j]);
    if (key > T) sel[atomicAdd(&s_count, 1u)] = ((uint64_t)(~key) << 32) | (uint32_t)j;
  }
  // equal to the threshold: the first `need` in index order
  for (int base = 0; base < p.n; base += nt) {
    if (s_running >= need) break;  // uniform: read after the barrier that ended the previous trip
    const int j = base + tid;
    const bool eq = j < p.n && orderable(sc[j]) == T;
    const uint64_t bal = __ballot(eq);
    if (lane == 0) s_wave[wv] = (unsigned)__popcll(bal);
    __syncthreads();
    unsigned before = s_running;
    for (int w = 0; w < wv; ++w) before += s_wave[w];
    const unsigned rank = before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
    if (eq && rank < need) sel[n_gt + rank] = ((uint64_t)(~T) << 32) | (uint32_t)j;
    __syncthreads();
    if (tid == 0) {
      unsigned tot = 0;
      for (int w = 0; w < (nt >> 6); ++w) tot += s_wave[w];
      s_running += tot;
    }
    __syncthreads();
  }
  __syncthreads();
  bitonic_sort_lds(sel, p.P);
  for (int j = tid; j < p.k; j += nt) p.out_idx[(int64_t)blockIdx.x * p.k + j] = (int32_t)(uint32_t)sel[j];
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
  int32_t layout;    // LAYOUT_ORIGINAL / LAYOUT_PARTITIONABLE
  int32_t E_total, env_offset;  // env e here is env env_offset + e of E_total (bh_set_env_window)
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
  const u32x2 ke = split_at(p.key0, p.key1, (uint32_t)(p.E_total + 1), (uint32_t)(1 + p.env_offset + e), p.layout);
  const int32_t *perm = shuffle_lds(ke, p.n_pairs, p.P_pairs, p.rounds, sk, x, y, p.layout);

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

// `PairScores._convert_actions` (breedgym/vector/vec_wrappers.py:100-112) behind the top-n of the n x n pair scores,
// one workgroup per env:
//     vals  = scores[best]                         the n best pair scores, best first
//     sm    = softmax(vals)                        f32; the one order-dependent step -- the sum -- in f64, rounded once
//     count = ceil(sm * n)                         offspring per pair
//     out   = repeat(pairs[best], count)[:n]       slot s takes the first pair whose cumulative count exceeds s
// (the reference's jnp.repeat(..., total_repeat_length=n)); pair = (best // n, best % n).
struct PairScoreParams {
  const float *scores;  // (E, n * n)
  const int32_t *best;  // (E, n) flat indices of the n best, best first (bh_topk)
  int32_t *actions;     // (E, n, 2)
  int32_t n;
  int64_t row_len;  // scores per env: n * n for the documented (E, n, n) action; the reference flattens whatever it is given
};

__global__ __launch_bounds__(1024) void pair_score_actions_kernel(const PairScoreParams p) {
  extern __shared__ __align__(16) float ps_lds[];  // n f32 (exp), n int32 (cumulative counts)
  __shared__ double part[16];
  __shared__ int carry;
  float *ex = ps_lds;
  int32_t *ends = reinterpret_cast<int32_t *>(ps_lds + p.n);
  const int e = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, n = p.n;
  const float *sc = p.scores + (int64_t)e * p.row_len;
  const int32_t *best = p.best + (int64_t)e * n;
  const float vmax = sc[best[0]];  // best first: the largest score
  double local = 0.0;
  for (int i = tid; i < n; i += nt) {
    const float x = expf(sc[best[i]] - vmax);
    ex[i] = x;
    local += (double)x;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
  if (lane == 0) part[wv] = local;
  if (tid == 0) carry = 0;
  __syncthreads();
  double total = 0.0;
  for (int w = 0; w < (nt >> 6); ++w) total += part[w];
  const float denom = (float)total;
  // inclusive cumulative counts, a block of `nt` pairs at a time
  for (int base = 0; base < n; base += nt) {
    const int i = base + tid;
    const int c = i < n ? (int)ceilf(ex[i] / denom * (float)n) : 0;
    int v = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(v, off);
      if (lane >= off) v += t;
    }
    __syncthreads();  // (part is free again; carry of the previous block is final)
    int *wsum = reinterpret_cast<int *>(part);
    if (lane == 63) wsum[wv] = v;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wv; ++w) before += wsum[w];
    if (i < n) ends[i] = before + v;
    __syncthreads();
    if (tid == nt - 1) carry = before + v;
  }
  __syncthreads();
  for (int s = tid; s < n; s += nt) {  // first pair whose cumulative count exceeds s
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (ends[mid] > s)
        hi = mid;
      else
        lo = mid + 1;
    }
    const int b = best[lo];
    p.actions[((int64_t)e * n + s) * 2] = b / n;
    p.actions[((int64_t)e * n + s) * 2 + 1] = b % n;
  }
}

}  // namespace bh
