// Population sharding (the role of the reference's per-device fan-out, breedgym/vector/vec_env.py:150-236,
// for ONE population cut row-wise over the ranks): where the winners of a global top-k live and where they land.
//
// Every rank holds the same list of winners (global row indices, best first).  Rank r owns rows
// [r n_local, (r + 1) n_local); the winners' rows travel by one all-gather of equally sized blocks, rank r's block
// holding the winners it owns in list order.  This kernel derives, from the list alone:
//   counts[r]    winners rank r owns (r < world), counts[world] = the largest of them
//   send_idx[j]  local row of the j-th winner THIS rank owns (j < cap; beyond its count: row 0 again, padding)
//   pos[i]       row of winner i in the gathered buffer of world x cap rows: owner(i) * cap + (winners of the same
//                owner earlier in the list)
// -- what distributed.py did with a one-hot matrix, a cumulative sum, a boolean-mask select and two host round trips.
// One workgroup walks the list in order, 1024 winners at a time: a wave ranks its lanes among equal owners with
// ballots, the waves hand each other their counts through LDS.  A winner that does not fit its owner's `cap` rows is
// counted but has no valid place: the caller compares counts[world] with cap (speculative block size, distributed.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bh {

constexpr int kMaxShardRanks = 256;

struct WinnerLayoutParams {
  const int32_t *winners;  // (k) global row indices
  int32_t *send_idx;       // (cap)
  int32_t *pos;            // (k)
  int32_t *counts;         // (world + 1)
  int32_t k, n_local, world, rank, cap;
};

__global__ __launch_bounds__(1024) void winner_layout_kernel(const WinnerLayoutParams p) {
  extern __shared__ int32_t wl_lds[];  // base[world], then per wave world counts
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6, W = p.world;
  int32_t *base = wl_lds, *wcnt = wl_lds + W;
  for (int i = tid; i < W; i += blockDim.x) base[i] = 0;
  for (int c0 = 0; c0 < p.k; c0 += blockDim.x) {
    for (int i = tid; i < nw * W; i += blockDim.x) wcnt[i] = 0;
    __syncthreads();
    const int i = c0 + tid;
    const bool in = i < p.k;
    int g = in ? p.winners[i] : 0;
    int owner = in ? min(max(g / p.n_local, 0), W - 1) : -1;
    // rank among the wave's lanes of the same owner (the lanes are in list order)
    int before = 0;
    uint64_t todo = __ballot(in);
    while (todo) {  // one turn per distinct owner in the wave (wave-uniform loop)
      const int o = __builtin_amdgcn_readfirstlane(__shfl(owner, __ffsll((long long)todo) - 1));
      const uint64_t same = __ballot(owner == o);
      if (owner == o) {
        before = __popcll(same & ((1ull << lane) - 1ull));
        if (before == 0) wcnt[wave * W + o] = __popcll(same);
      }
      todo &= ~same;
    }
    __syncthreads();
    if (in) {
      int within = base[owner] + before;
      for (int w = 0; w < wave; ++w) within += wcnt[w * W + owner];
      p.pos[i] = owner * p.cap + min(within, p.cap - 1);
      if (owner == p.rank && within < p.cap) p.send_idx[within] = g - p.rank * p.n_local;
    }
    __syncthreads();
    for (int o = tid; o < W; o += blockDim.x) {
      int s = base[o];
      for (int w = 0; w < nw; ++w) s += wcnt[w * W + o];
      base[o] = s;
    }
    __syncthreads();
  }
  __syncthreads();
  for (int o = tid; o < W; o += blockDim.x) p.counts[o] = base[o];
  if (tid == 0) {
    int m = 0;
    for (int o = 0; o < W; ++o) m = max(m, base[o]);
    p.counts[W] = m;
  }
  for (int j = min(base[p.rank], p.cap) + tid; j < p.cap; j += blockDim.x) p.send_idx[j] = 0;
}

// idx_out[i] = pos[pairs[i]]: parent indices into the list of winners -> rows of the gathered buffer.  A pair index
// follows JAX's retrieval rule on the k winners (negative wraps once, then clamps), as the cross launches apply it to
// a pool of k rows.
__global__ __launch_bounds__(256) void remap_pairs_kernel(const int32_t *__restrict__ pos, int32_t k,
                                                         const int32_t *__restrict__ pairs, int64_t n,
                                                         int32_t *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int j = pairs[i];
    if (j < 0) j += k;
    j = min(max(j, 0), k - 1);
    out[i] = pos[j];
  }
}

}  // namespace bh
