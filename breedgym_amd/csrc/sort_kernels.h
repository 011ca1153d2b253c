// Stable key-value sort in HBM over many workgroups: what JAX's sort-based shuffle needs once a draw no longer
// fits one workgroup's LDS.
//
// jax.random.permutation / choice(replace=False) (jax/_src/random.py `_shuffle`) = `rounds` times
//     key, sub = split(key);  x = stable_sort_by_key(random_bits(sub, 32, N), x)
// which the reference reaches through `_random_selection` (breedgym/vector/vec_env.py:22-27: the founder draw of
// every env at reset) and `SelectionScores._convert_actions` (breedgym/vector/vec_wrappers.py:61-79: the random
// subset of the k(k-1)/2 diallel pairs).  Small draws stay inside LDS (founders.h, select_kernels.h); here the
// same rounds run as an LSD radix sort on the 32-bit Threefry word -- four passes of 8 bits, each
//     sort_hist_kernel     per tile and digit value: how many of the tile's keys carry it
//     sort_scan_kernel     per digit value: exclusive prefix over the tiles, and the value's total
//     sort_scatter_kernel  every element to  base(value) + prefix(value, tile) + its rank among the tile's
//                          elements of the same value, in index order
// so equal words keep their order: the position is the tiebreak, as in the LDS kernels' (word << 32 | position).
// Segments (envs) are sorted side by side: blockIdx.y is the segment.  HBM-bound: a pass reads 16 and writes
// 8 bytes per element (keys and values travel as one 8-byte word).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "threefry.h"

namespace bh {

constexpr int kSortThreads = 512;                      // 8 waves
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kSortItems = 16;                         // elements per lane
constexpr int kSortTile = kSortThreads * kSortItems;   // 8192 elements per workgroup and pass
constexpr int kSortWaveSpan = 64 * kSortItems;         // consecutive elements one wave ranks

// an element: .x the 32-bit key, .y the value carried along -- one 8-byte word, so that the scatter of a pass is one
// store per element (two 4-byte stores to two arrays ran at half the rate: profiles/r03_sort_pairs.txt)
struct SortParams {
  const uint2 *in;
  uint2 *out;
  uint32_t *hist;       // (segs, 256, nb): counts, then exclusive prefixes over the tiles
  uint32_t *bin_total;  // (segs, 256)
  int64_t seg_stride;   // elements from one segment to the next
  int32_t N, nb, shift;
};

__global__ __launch_bounds__(kSortThreads) void sort_hist_kernel(const SortParams p) {
  __shared__ unsigned cnt[kSortWaves][256];  // a count per wave: eight times fewer lanes meet in a counter
  const int tid = threadIdx.x, wv = tid >> 6, tile = blockIdx.x, seg = blockIdx.y;
  for (int j = tid; j < kSortWaves * 256; j += kSortThreads) (&cnt[0][0])[j] = 0;
  __syncthreads();
  const uint2 *in = p.in + (int64_t)seg * p.seg_stride;
  const int base = tile * kSortTile;
  uint32_t key[kSortItems];
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {  // every lane loads (an element past the end: the last one, not counted)
    const int i = base + j * kSortThreads + tid;
    key[j] = in[i < p.N ? i : p.N - 1].x;
  }
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const int i = base + j * kSortThreads + tid;
    if (i < p.N) atomicAdd(&cnt[wv][(key[j] >> p.shift) & 255u], 1u);
  }
  __syncthreads();
  if (tid < 256) {
    unsigned sum = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) sum += cnt[w][tid];
    p.hist[((int64_t)seg * 256 + tid) * p.nb + tile] = sum;
  }
}

// inclusive scan over the 256 threads of a workgroup; returns this thread's inclusive value, *total the sum
__device__ __forceinline__ unsigned scan256(unsigned v, unsigned *wsum, unsigned *total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned t = __shfl_up(v, off);
    if (lane >= off) v += t;
  }
  if (lane == 63) wsum[wv] = v;
  __syncthreads();
  unsigned before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const unsigned s = wsum[w];
    if (w < wv) before += s;
    all += s;
  }
  __syncthreads();
  *total = all;
  return v + before;
}

// one workgroup per (digit value, segment)
__global__ __launch_bounds__(256) void sort_scan_kernel(const SortParams p) {
  __shared__ unsigned wsum[4];
  const int tid = threadIdx.x, d = blockIdx.x, seg = blockIdx.y;
  uint32_t *h = p.hist + ((int64_t)seg * 256 + d) * p.nb;
  unsigned carry = 0;
  for (int t0 = 0; t0 < p.nb; t0 += 256) {
    const int t = t0 + tid;
    const unsigned v = t < p.nb ? h[t] : 0u;
    unsigned total;
    const unsigned inc = scan256(v, wsum, &total);
    if (t < p.nb) h[t] = carry + inc - v;
    carry += total;
  }
  if (tid == 0) p.bin_total[seg * 256 + d] = carry;
}

__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(const SortParams p) {
  __shared__ unsigned cnt[kSortWaves][256];  // per wave: running count, then the wave's first slot, per digit value
  __shared__ unsigned gbase[256];
  __shared__ unsigned wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, tile = blockIdx.x, seg = blockIdx.y;
  for (int j = tid; j < kSortWaves * 256; j += kSortThreads) (&cnt[0][0])[j] = 0;
  // where this tile's elements of every digit value start: all smaller values, then the same value in earlier
  // tiles.  Threads 0-255 own a digit value each; the other four waves only walk through the barriers.
  {
    const unsigned v = tid < 256 ? p.bin_total[seg * 256 + tid] : 0u;
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned t = __shfl_up(inc, off);
      if (lane >= off) inc += t;
    }
    if (lane == 63 && wv < 4) wsum[wv] = inc;
    __syncthreads();
    if (tid < 256) {
      unsigned before = 0;
      for (int w = 0; w < wv; ++w) before += wsum[w];
      gbase[tid] = before + inc - v + p.hist[((int64_t)seg * 256 + tid) * p.nb + tile];
    }
  }
  __syncthreads();

  const uint2 *in = p.in + (int64_t)seg * p.seg_stride;
  const int wbase = tile * kSortTile + wv * kSortWaveSpan;
  uint2 e[kSortItems];
  unsigned lr[kSortItems];
  const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const int i = wbase + j * 64 + lane;
    e[j] = i < p.N ? in[i] : make_uint2(0u, 0u);
  }
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const int i = wbase + j * 64 + lane;
    const bool valid = i < p.N;
    const unsigned d = (e[j].x >> p.shift) & 255u;
    uint64_t peers = __ballot(valid);  // the lanes of this trip that carry the same digit value
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const unsigned rank = (unsigned)__popcll(peers & lt);
    // LDS operations of one wave are served in issue order: every peer reads the count before the first peer bumps it
    const unsigned pre = cnt[wv][d];
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0) cnt[wv][d] = pre + (unsigned)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    lr[j] = pre + rank;
  }
  __syncthreads();
  if (tid < 256) {  // counts -> first slots, wave after wave (= index order)
    unsigned run = gbase[tid];
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) {
      const unsigned c = cnt[w][tid];
      cnt[w][tid] = run;
      run += c;
    }
  }
  __syncthreads();
  uint2 *out = p.out + (int64_t)seg * p.seg_stride;
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const int i = wbase + j * 64 + lane;
    if (i < p.N) out[cnt[wv][(e[j].x >> p.shift) & 255u] + lr[j]] = e[j];
  }
}

// ---- what is sorted ---------------------------------------------------------------------------------------
// The shuffle keys of every round of every segment: segment s is env env_offset + seg0 + s of E_total, its key
// split(env_key, E_total + 1)[1 + env]; round r sorts by random_bits(sub_r) with (key, sub_r) = split(key).
struct ShuffleKeyParams {
  uint32_t *subkeys;  // (segs, rounds, 2)
  uint32_t key0, key1;
  int32_t segs, rounds, layout;
  int32_t E_total, env_first;  // env of segment 0
};

__global__ void shuffle_subkeys_kernel(const ShuffleKeyParams p) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= p.segs) return;
  u32x2 k = split_at(p.key0, p.key1, (uint32_t)(p.E_total + 1), (uint32_t)(1 + p.env_first + s), p.layout);
  for (int rd = 0; rd < p.rounds; ++rd) {
    const u32x2 nk = split_at(k.x, k.y, 2u, 0u, p.layout), sub = split_at(k.x, k.y, 2u, 1u, p.layout);
    k = nk;
    p.subkeys[((int64_t)s * p.rounds + rd) * 2] = sub.x;
    p.subkeys[((int64_t)s * p.rounds + rd) * 2 + 1] = sub.y;
  }
}

struct ShuffleGenParams {
  const uint32_t *subkeys;  // (segs, rounds, 2)
  uint2 *kv;                // (segs, seg_stride): .x = random_bits(sub, N), .y = 0..N-1 when `init`, else kept
  int64_t seg_stride;
  int32_t N, rounds, round, layout, init;
};

// random_bits(sub, N) in either layout (threefry.h): the original one hashes positions (i, i + h) together
__global__ __launch_bounds__(256) void shuffle_gen_kernel(const ShuffleGenParams p) {
  const int seg = blockIdx.y;
  const uint32_t s0 = p.subkeys[((int64_t)seg * p.rounds + p.round) * 2],
                 s1 = p.subkeys[((int64_t)seg * p.rounds + p.round) * 2 + 1];
  uint2 *kv = p.kv + (int64_t)seg * p.seg_stride;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (p.layout == LAYOUT_PARTITIONABLE) {
    if (i < p.N) {
      kv[i].x = bits_word_partitionable(s0, s1, (uint32_t)i);
      if (p.init) kv[i].y = (uint32_t)i;
    }
  } else {
    const int h = (p.N + 1) >> 1;
    if (i < h) {
      const bool second = i + h < p.N;  // odd N: the last pair is padded with a 0 counter
      const u32x2 o = threefry2x32(s0, s1, (uint32_t)i, second ? (uint32_t)(i + h) : 0u);
      kv[i].x = o.x;
      if (p.init) kv[i].y = (uint32_t)i;
      if (second) {
        kv[i + h].x = o.y;
        if (p.init) kv[i + h].y = (uint32_t)(i + h);
      }
    }
  }
}

// keys for jax.lax.top_k as a sort: ascending ~orderable(score) = descending score, ties to the lower index
struct ScoreKeyParams {
  const float *scores;  // (segs, N)
  uint2 *kv;
  int64_t seg_stride;
  int32_t N;
};

__global__ __launch_bounds__(256) void score_keys_kernel(const ScoreKeyParams p) {
  const int seg = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.N) return;
  const uint32_t u = __float_as_uint(p.scores[(int64_t)seg * p.N + i]);
  p.kv[(int64_t)seg * p.seg_stride + i] = make_uint2(~(u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u)), (uint32_t)i);
}

// out[s, 0:take] = the values of the first `take` elements of segment s
__global__ __launch_bounds__(256) void take_front_kernel(const uint2 *kv, int64_t seg_stride, int32_t take, int32_t *out) {
  const int seg = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i < take) out[(int64_t)seg * take + i] = (int32_t)kv[(int64_t)seg * seg_stride + i].y;
}

// `SelectionScores._convert_actions`' tail (vec_wrappers.py:66-79) from a shuffled list of pair ranks:
// slot s of env e takes pair perm[s / n_offspring]; rank t -> (i, j), i < j, row-major upper triangle of k x k.
struct PairActionParams {
  const uint2 *perm;    // (segs, seg_stride) shuffled elements, .y = pair rank (the first n_crosses are read)
  const int32_t *best;  // (segs, k) top-k indices, best first
  int32_t *actions;     // (segs, n, 2)
  int64_t seg_stride;
  int32_t n, k, n_offspring;
};

__global__ __launch_bounds__(256) void pair_actions_kernel(const PairActionParams p) {
  const int seg = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
  if (s >= p.n) return;
  const int64_t t = p.perm[(int64_t)seg * p.seg_stride + s / p.n_offspring].y, k = p.k;
  // rows before row i hold i (2k - i - 1) / 2 pairs
  const double b = 2.0 * (double)k - 1.0;
  int64_t i = (int64_t)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
  i = i < 0 ? 0 : (i > k - 2 ? k - 2 : i);
  while (i > 0 && i * (2 * k - i - 1) / 2 > t) --i;
  while ((i + 1) * (2 * k - i - 2) / 2 <= t) ++i;
  const int64_t j = t - i * (2 * k - i - 1) / 2 + i + 1;
  const int32_t *best = p.best + (int64_t)seg * p.k;
  p.actions[((int64_t)seg * p.n + s) * 2] = best[i];
  p.actions[((int64_t)seg * p.n + s) * 2 + 1] = best[j];
}

}  // namespace bh
