// JAX's sort-based shuffle when only a FRONT of the result is wanted -- which is every caller this library has:
//   `_random_selection` (breedgym/vector/vec_env.py:22-27)            choice(key, germplasm, (n,), replace=False)
//   `SelectionScores._convert_actions` (vec_wrappers.py:61-79)        choice(key, k(k-1)/2 pairs, (n_crosses,), replace=False)
// = permutation(key, N)[:take], and jax's `_shuffle` is `rounds` times
//       key, sub = split(key);  x = stable_sort_by_key(random_bits(sub, 32, N), x),     x = arange(N) at the start.
// Round 3 sorted all N elements `rounds` times (sort_kernels.h: 21.5 M pair ranks x 3 rounds x 4 radix passes for the
// 8,192 pairs BASELINE configs[3] asks for).  Here nothing is sorted.  With s_r = the stable argsort of round r's
// words, x_R[j] = s_1[s_2[... s_R[j]]]: the wanted set is propagated BACKWARDS --
//       want_R = {0 .. take-1};   p_r[j] = the position whose (word_r, position) has rank want_r[j];   want_(r-1) = p_r
// and x[j] = p_1[j].  A round's words are recomputable from the position (Threefry), so no key array exists either.
// Per round, five small launches:
//   sparse_hist      every position's word -> a histogram of its top 16 bits.  A workgroup takes <= 65,535 positions
//                    and counts in LDS (2^16 16-bit counters: they cannot overflow), then writes its counts out: global
//                    atomics were measured at 27 G/s for scattered addresses (tools/atomic_rate.hip: 790 us for the
//                    21.5 M words), the LDS histogram with its partial rows written at 26 us.
//   sparse_reduce    per bucket: the total, and in place of every workgroup's count the count of the workgroups before it;
//                    the sums of the totals over groups of 16 buckets, and the totals' prefix inside a group
//   sparse_map       every wanted rank -> its bucket (a binary search over the 4096 group prefixes in LDS, then the
//                    group's one line of 16 prefixes) and its rank inside; the buckets that hold a wanted rank are marked
//   sparse_gather    the words again; an element of a marked bucket goes to  start(bucket) + (counts of the workgroups
//                    before) + (its turn in this workgroup: an LDS atomic on the row sparse_reduce left) -- every element
//                    has its own slot, no global atomic.  N/2^16 elements per wanted rank are moved (328 at 21.5 M).
//   sparse_select    a wave per wanted rank: the element of that rank in its bucket's list, ties to the lower position
//                    (a 64-bin histogram on the next six bits narrows the bucket to a handful of candidates first)
// Exactly JAX's result: the order (word, position) is the order of a stable sort by word.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "threefry.h"

namespace bh {

constexpr int kSpBuckets = 65536;     // the top 16 bits of a round's word
constexpr int kSpThreads = 1024;
constexpr int kSpMaxPositions = 65535;  // per workgroup of sparse_hist: up to here a 16-bit counter holds them all, whatever the words
constexpr int kSpMaxBucket = 32768;     // elements of one bucket beyond which the call gives up (device error word):
                                        // the mean is N / 65536 <= 512; 2^15 equal top halves among Threefry words of
                                        // distinct counters do not happen
constexpr int kSpGroups = kSpBuckets / 16;
constexpr int kSpLdsWords = kSpBuckets / 2;  // two 16-bit counters per LDS word
constexpr uint32_t kSpErrBucket = 0x53500001u;  // written to the error word: a bucket over kSpMaxBucket

struct SparseParams {
  const uint32_t *subkeys;  // (segs, rounds, 2)  shuffle_subkeys_kernel
  uint16_t *partial;        // (segs, nchunk_pad, 65536): nchunk rows are used, the rest is room sparse_reduce may read
  uint32_t *total;          // (segs, 65536)
  uint32_t *pre16;          // (segs, 65536): the totals of the buckets of the same group of 16 before this one
  uint32_t *group_sum;      // (segs, 4096): the totals of the groups of 16 buckets
  uint32_t *marks;          // (segs, 2048): bit d = bucket d holds a wanted rank
  const uint32_t *want;     // (segs, take) wanted ranks, or null: 0 .. take-1
  uint32_t *wbucket, *wrank, *wstart;  // (segs, take): bucket, rank inside it, where the bucket starts in the list
  uint2 *list;              // (segs, list_stride >= N) elements (word, position) of the marked buckets, each bucket at
                            // the place a full sort by bucket would give it (most of the array is never touched)
  uint2 *out;               // (segs, out_stride): the element of every wanted rank, in the order of `want`
  uint32_t *next_want;      // (segs, take): their positions = the wanted ranks of the round before
  uint32_t *err;            // the context's device-side error word
  int64_t list_stride, out_stride;
  int32_t N, take, nchunk, nchunk_pad, chunk_items, rounds, round, layout;
};

// The words of items [i0, i1) of a round, item by item: f(word, position).  Original layout: item i < ceil(N/2) is the
// Threefry call on counters (i, i + h), whose two outputs are the words of positions i and i + h (odd N: the last
// call's second counter is 0 and its second output unused); partitionable: item = position.
// Two items per lane and turn (kSpThreads apart) through the hand-placed two-chain block of threefry.h, the round's key
// schedule in scalar registers: these sweeps are Threefry and little else (two per round over every position of the
// list: 21.5 M at BASELINE configs[3]'s selection).  A lane whose second item lies past the chunk redoes its first.
// The order in which f is called does not matter to either caller (counts; slots inside a bucket).
template <class F>
__device__ __forceinline__ void sparse_words(uint32_t s0, uint32_t s1, int layout, int N, int i0, int i1, F f) {
  const uint32_t k0 = __builtin_amdgcn_readfirstlane(s0), k1 = __builtin_amdgcn_readfirstlane(s1);
  const KeySched ks = key_sched(k0, k1);
  if (layout == LAYOUT_PARTITIONABLE) {
    for (int i = i0 + (int)threadIdx.x; i < i1; i += 2 * kSpThreads) {
      const int ib = i + kSpThreads;
      const bool two = ib < i1;
      uint32_t a0 = k0, a1 = (uint32_t)i + k1, b0 = k0, b1 = (uint32_t)(two ? ib : i) + k1;
      threefry2x32_x2(ks, a0, a1, b0, b1);
      f(a0 ^ a1, i);
      if (two) f(b0 ^ b1, ib);
    }
  } else {
    const int h = (N + 1) >> 1;
    for (int i = i0 + (int)threadIdx.x; i < i1; i += 2 * kSpThreads) {
      const int ib = i + kSpThreads;
      const bool two = ib < i1;
      const int jb = two ? ib : i;
      const bool sa = i + h < N, sb = jb + h < N;
      uint32_t a0 = (uint32_t)i + k0, a1 = (sa ? (uint32_t)(i + h) : 0u) + k1;
      uint32_t b0 = (uint32_t)jb + k0, b1 = (sb ? (uint32_t)(jb + h) : 0u) + k1;
      threefry2x32_x2(ks, a0, a1, b0, b1);
      f(a0, i);
      if (sa) f(a1, i + h);
      if (two) {
        f(b0, ib);
        if (sb) f(b1, ib + h);
      }
    }
  }
}
__host__ __device__ __forceinline__ int sparse_items(int layout, int N) { return layout == LAYOUT_PARTITIONABLE ? N : (N + 1) >> 1; }

__global__ __launch_bounds__(kSpThreads) void sparse_hist_kernel(const SparseParams p) {
  extern __shared__ uint32_t sp_lds[];  // kSpLdsWords
  const int tid = threadIdx.x, chunk = blockIdx.x, seg = blockIdx.y;
  for (int j = tid; j < kSpLdsWords; j += kSpThreads) sp_lds[j] = 0;
  __syncthreads();
  const uint32_t s0 = p.subkeys[((int64_t)seg * p.rounds + p.round) * 2], s1 = p.subkeys[((int64_t)seg * p.rounds + p.round) * 2 + 1];
  const int items = sparse_items(p.layout, p.N), i0 = chunk * p.chunk_items, i1 = min(i0 + p.chunk_items, items);
  // a workgroup of up to 65,535 positions cannot wrap a 16-bit count; a longer chunk (one round of workgroups instead of
  // two halves every pass over the count rows: select.hip, sparse_geom) could -- only with more than 65,535 equal top
  // halves among its words, which is the error sparse_select reports too, seen here by the counter's value before
  if (2 * (i1 - i0) <= kSpMaxPositions || (p.layout == LAYOUT_PARTITIONABLE && i1 - i0 <= kSpMaxPositions)) {
    sparse_words(s0, s1, p.layout, p.N, i0, i1, [&](uint32_t w, int) {
      const uint32_t d = w >> 16;
      atomicAdd(&sp_lds[d >> 1], 1u << (16 * (d & 1)));
    });
  } else {
    bool full = false;
    sparse_words(s0, s1, p.layout, p.N, i0, i1, [&](uint32_t w, int) {
      const uint32_t d = w >> 16;
      const uint32_t old = atomicAdd(&sp_lds[d >> 1], 1u << (16 * (d & 1)));
      full |= ((old >> (16 * (d & 1))) & 0xFFFFu) == 0xFFFFu;
    });
    if (full) *p.err = kSpErrBucket;
  }
  __syncthreads();
  uint4 *dst = reinterpret_cast<uint4 *>(p.partial + ((int64_t)seg * p.nchunk_pad + chunk) * kSpBuckets);
  for (int j = tid; j < kSpLdsWords / 4; j += kSpThreads) dst[j] = reinterpret_cast<const uint4 *>(sp_lds)[j];
}

// A workgroup of sixteen waves per block of 256 buckets, four buckets per lane, the workgroups' rows dealt to the waves
// in sixteen runs of `rows_per_wave` (a multiple of BATCH; the array is padded to 16 runs): partial[c][d] <- sum of
// partial[c' < c][d], total[d] <- the sum over all c; group_sum = the totals of every 16 buckets, pre16[d] = the
// totals of d's group before d; the block's marks are cleared for sparse_map.  A wave asks for BATCH rows at a time
// (BATCH x 512 bytes in flight per wave: the pass is 2 x 64 MB at 21.5 M elements and nothing but latency to hide),
// sums its run, and reads it a second time (out of the caches) once the waves before it have told their sums.
template <int BATCH>
__global__ __launch_bounds__(1024) void sparse_reduce_kernel(const SparseParams p) {
  __shared__ uint32_t wtot[16][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, blk = blockIdx.x, seg = blockIdx.y, d0 = blk * 256 + 4 * lane;
  const int rpw = p.nchunk_pad / 16, c0 = wv * rpw;
  constexpr int64_t kRow = kSpBuckets / 4;  // uint2 per row
  uint2 *col = reinterpret_cast<uint2 *>(p.partial + ((int64_t)seg * p.nchunk_pad + c0) * kSpBuckets + d0);
  uint32_t sum[4] = {0u, 0u, 0u, 0u};
  for (int b0 = 0; b0 < rpw; b0 += BATCH) {
    uint2 v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) v[u] = col[(b0 + u) * kRow];  // every load issued, none under a condition
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      if (c0 + b0 + u >= p.nchunk) v[u] = make_uint2(0u, 0u);  // padding rows hold anything
      sum[0] += v[u].x & 0xFFFFu;
      sum[1] += v[u].x >> 16;
      sum[2] += v[u].y & 0xFFFFu;
      sum[3] += v[u].y >> 16;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) wtot[wv][4 * lane + q] = sum[q];
  __syncthreads();
  uint32_t run[4] = {0u, 0u, 0u, 0u}, all[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int w = 0; w < 16; ++w)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t t = wtot[w][4 * lane + q];
      if (w < wv) run[q] += t;
      all[q] += t;
    }
  for (int b0 = 0; b0 < rpw; b0 += BATCH) {
    uint2 v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) v[u] = col[(b0 + u) * kRow];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      if (c0 + b0 + u < p.nchunk) {  // (a running count stays below 2^16 unless the bucket is over the limit sparse_select reports)
        col[(b0 + u) * kRow] = make_uint2((run[0] & 0xFFFFu) | (run[1] << 16), (run[2] & 0xFFFFu) | (run[3] << 16));
        run[0] += v[u].x & 0xFFFFu;
        run[1] += v[u].x >> 16;
        run[2] += v[u].y & 0xFFFFu;
        run[3] += v[u].y >> 16;
      }
    }
  }
  if (wv != 0) return;
  const uint32_t mine = all[0] + all[1] + all[2] + all[3];
  uint32_t g = mine;  // inclusive over the four lanes of a group of 16 buckets
  const uint32_t up1 = __shfl_up(g, 1);
  if (lane & 3) g += up1;
  const uint32_t up2 = __shfl_up(g, 2);
  if ((lane & 3) >= 2) g += up2;
  const uint32_t before = g - mine;
  *reinterpret_cast<uint4 *>(p.total + (int64_t)seg * kSpBuckets + d0) = make_uint4(all[0], all[1], all[2], all[3]);
  *reinterpret_cast<uint4 *>(p.pre16 + (int64_t)seg * kSpBuckets + d0) =
      make_uint4(before, before + all[0], before + all[0] + all[1], before + all[0] + all[1] + all[2]);
  if ((lane & 3) == 3) p.group_sum[(int64_t)seg * kSpGroups + blk * 16 + (lane >> 2)] = g;
  if (lane < 8) p.marks[(int64_t)seg * (kSpBuckets / 32) + blk * 8 + lane] = 0u;
}

// exclusive prefix of a segment's 4096 group sums into LDS (base[4096] = N); 256 threads or 1024
__device__ __forceinline__ void sparse_group_bases(const SparseParams &p, int seg, uint32_t *base, uint32_t *wsum) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nt = blockDim.x, per = kSpGroups / nt;  // 16 or 4 per thread
  const uint32_t *gs = p.group_sum + (int64_t)seg * kSpGroups + tid * per;
  uint32_t v[16], mine = 0;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    v[u] = u < per ? gs[u] : 0u;
    mine += v[u];
  }
  uint32_t inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint32_t run = inc - mine;
  for (int w = 0; w < wv; ++w) run += wsum[w];
#pragma unroll
  for (int u = 0; u < 16; ++u)
    if (u < per) {
      base[tid * per + u] = run;
      run += v[u];
    }
  if (tid == nt - 1) base[kSpGroups] = run;
  __syncthreads();
}

// a thread per wanted rank
__global__ __launch_bounds__(256) void sparse_map_kernel(const SparseParams p) {
  __shared__ uint32_t base[kSpGroups + 1];
  __shared__ uint32_t wsum[16];
  const int seg = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  sparse_group_bases(p, seg, base, wsum);
  if (j >= p.take) return;
  const uint32_t rho = p.want ? p.want[(int64_t)seg * p.take + j] : (uint32_t)j;
  // the LAST group whose prefix is <= rho (empty groups share their prefix with the group behind them)
  int lo = 0, hi = kSpGroups;  // base[lo] <= rho < base[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (base[mid] <= rho)
      lo = mid;
    else
      hi = mid;
  }
  // ... and the last bucket of that group: its 16 prefixes are one 64-byte line, the only trip to memory
  const uint4 *pl = reinterpret_cast<const uint4 *>(p.pre16 + (int64_t)seg * kSpBuckets + lo * 16);
  const uint4 q0 = pl[0], q1 = pl[1], q2 = pl[2], q3 = pl[3];
  const uint32_t pre[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
  const uint32_t r = rho - base[lo];
  int a = 0;
  uint32_t pa = 0;
#pragma unroll
  for (int u = 1; u < 16; ++u)
    if (pre[u] <= r) {
      a = u;
      pa = pre[u];
    }
  const uint32_t d = (uint32_t)(lo * 16 + a);
  p.wbucket[(int64_t)seg * p.take + j] = d;
  p.wrank[(int64_t)seg * p.take + j] = r - pa;
  p.wstart[(int64_t)seg * p.take + j] = base[lo] + pa;
  // neighbours in j often want the same bucket (the last round's wanted ranks are 0 .. take-1: hundreds of them per
  // bucket, and same-address atomics are served one after the other at the memory side): only the first of a run marks
  const uint32_t left = __shfl_up(d, 1);
  if ((threadIdx.x & 63) == 0 || left != d) atomicOr(&p.marks[(int64_t)seg * (kSpBuckets / 32) + (d >> 5)], 1u << (d & 31));
}

__global__ __launch_bounds__(kSpThreads) void sparse_gather_kernel(const SparseParams p) {
  extern __shared__ uint32_t sp_lds[];  // kSpLdsWords of running slots (this workgroup's row of `partial`), then the marks
  __shared__ uint32_t base[kSpGroups + 1];
  __shared__ uint32_t wsum[16];
  uint32_t *marks = sp_lds + kSpLdsWords;
  const int tid = threadIdx.x, chunk = blockIdx.x, seg = blockIdx.y;
  sparse_group_bases(p, seg, base, wsum);
  const uint4 *row = reinterpret_cast<const uint4 *>(p.partial + ((int64_t)seg * p.nchunk_pad + chunk) * kSpBuckets);
  for (int j = tid; j < kSpLdsWords / 4; j += kSpThreads) reinterpret_cast<uint4 *>(sp_lds)[j] = row[j];
  for (int j = tid; j < kSpBuckets / 32; j += kSpThreads) marks[j] = p.marks[(int64_t)seg * (kSpBuckets / 32) + j];
  __syncthreads();
  const uint32_t s0 = p.subkeys[((int64_t)seg * p.rounds + p.round) * 2], s1 = p.subkeys[((int64_t)seg * p.rounds + p.round) * 2 + 1];
  const uint32_t *pl = p.pre16 + (int64_t)seg * kSpBuckets;
  uint2 *list = p.list + (int64_t)seg * p.list_stride;
  const int items = sparse_items(p.layout, p.N), i0 = chunk * p.chunk_items, i1 = min(i0 + p.chunk_items, items);
  sparse_words(s0, s1, p.layout, p.N, i0, i1, [&](uint32_t w, int pos) {
    const uint32_t d = w >> 16;
    if ((marks[d >> 5] >> (d & 31)) & 1u) {
      const uint32_t old = atomicAdd(&sp_lds[d >> 1], 1u << (16 * (d & 1)));
      const uint32_t slot = (old >> (16 * (d & 1))) & 0xFFFFu;  // workgroups before + this workgroup's elements before
      list[base[d >> 4] + pl[d] + slot] = make_uint2(w, (uint32_t)pos);
    }
  });
}

__device__ __forceinline__ bool sparse_less(uint2 a, uint2 b) { return a.x < b.x || (a.x == b.x && a.y < b.y); }

// a wave per wanted rank; 4 waves per workgroup
__global__ __launch_bounds__(256) void sparse_select_kernel(const SparseParams p) {
  constexpr int kCand = 256;
  __shared__ uint32_t sub[4][64];
  __shared__ uint2 cand[4][kCand];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, seg = blockIdx.y;
  const int j = blockIdx.x * 4 + wv;
  if (j >= p.take) return;  // wave-uniform; no workgroup barrier below
  const uint32_t d = p.wbucket[(int64_t)seg * p.take + j], r = p.wrank[(int64_t)seg * p.take + j];
  const uint32_t total = p.total[(int64_t)seg * kSpBuckets + d];
  if (total > (uint32_t)kSpMaxBucket && lane == 0) *p.err = kSpErrBucket;  // the 16-bit running counts may have wrapped
  const int len = (int)min(total, (uint32_t)kSpMaxBucket);
  const uint2 *L = p.list + (int64_t)seg * p.list_stride + p.wstart[(int64_t)seg * p.take + j];
  if (len <= 64) {  // a handful of elements (N up to a few million): one each, everybody counts who is below
    const uint2 e = lane < len ? L[lane] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    uint32_t below = 0;
    for (int k = 0; k < len; ++k) {
      const uint2 o = make_uint2(__shfl(e.x, k), __shfl(e.y, k));
      below += sparse_less(o, e) ? 1u : 0u;
    }
    if (lane < len && below == r) {
      p.out[(int64_t)seg * p.out_stride + j] = e;
      p.next_want[(int64_t)seg * p.take + j] = e.y;
    }
    return;
  }
  // the next six bits of the word: which sixty-fourth of the bucket holds rank r
  sub[wv][lane] = 0;
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < len; i += 64) atomicAdd(&sub[wv][(L[i].x >> 10) & 63u], 1u);
  __builtin_amdgcn_wave_barrier();
  const uint32_t c = sub[wv][lane];
  uint32_t inc = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  const uint64_t hit = __ballot(inc - c <= r && r < inc);  // exactly one lane unless the lists are broken
  const int sb = hit ? __ffsll((long long)hit) - 1 : 0;
  const uint32_t r2 = r - __shfl(inc - c, sb);
  const int m = (int)__shfl(c, sb);
  uint2 found = make_uint2(0u, 0u);
  bool have = false;
  if (m <= kCand) {
    // the candidates of that part into LDS, in any order
    int filled = 0;
    for (int i0 = 0; i0 < len; i0 += 64) {
      const int i = i0 + lane;
      const uint2 e = i < len ? L[i] : make_uint2(0u, 0u);
      const bool is = i < len && ((e.x >> 10) & 63u) == (uint32_t)sb;
      const uint64_t b = __ballot(is);
      if (is) cand[wv][filled + __popcll(b & ((1ull << lane) - 1ull))] = e;
      filled += __popcll(b);
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < m; i += 64) {
      const uint2 e = cand[wv][i];
      uint32_t below = 0;
      for (int k = 0; k < m; ++k) below += sparse_less(cand[wv][k], e) ? 1u : 0u;
      if (below == r2) {
        found = e;
        have = true;
      }
    }
  } else {  // a part of more than kCand elements (never seen): count in place
    for (int i = lane; i < len; i += 64) {
      const uint2 e = L[i];
      if (((e.x >> 10) & 63u) != (uint32_t)sb) continue;
      uint32_t below = 0;
      for (int k = 0; k < len; ++k) {
        const uint2 o = L[k];
        below += (((o.x >> 10) & 63u) == (uint32_t)sb && sparse_less(o, e)) ? 1u : 0u;
      }
      if (below == r2) {
        found = e;
        have = true;
      }
    }
  }
  if (have) {  // one lane
    p.out[(int64_t)seg * p.out_stride + j] = found;
    p.next_want[(int64_t)seg * p.take + j] = found.y;
  }
}

}  // namespace bh
