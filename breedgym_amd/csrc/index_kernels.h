// Selection indices on the device: the optimal haploid of every individual and its value
// (reference breedgym/utils/index_functions.py:21-78, host NumPy loops there).
//
// One wave per individual.  Without blocks, marker m of the optimal haploid is any(alleles) where the
// first trait's effect is positive and all(alleles) elsewhere.  With blocks (B per chromosome), the
// haplotype with the larger block value  sum_m sum_t allele * effect[m, t]  (the first on ties, like
// argmax) supplies the block.  The value is 2 * sum_m OHP[m] * effect[m, t], accumulated in f64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bh {

constexpr int kOhvTraits = 8;  // traits per launch

struct OhvParams {
  const uint8_t *pop;          // (n_rows, pitch) byte-layout rows
  const double *eff;           // (T, effPitch) f64 effects, trait-major
  const double *effsum;        // (M) sum over traits, for the block values
  const int32_t *block_start;  // (n_blocks + 1) device, or null
  uint8_t *ohp;                // (n_rows, M) or null
  float *ohv;                  // (n_rows, T) or null
  int64_t pitch, effPitch;
  int32_t M, T, t0, tcount, n_rows, n_blocks;
};

__device__ __forceinline__ double wave_sum(double s) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  return s;
}

__global__ __launch_bounds__(256) void optimal_haploid_kernel(const OhvParams p) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= p.n_rows) return;
  const uint8_t *g = p.pop + (int64_t)row * p.pitch;
  uint8_t *h = (p.ohp && p.t0 == 0) ? p.ohp + (int64_t)row * p.M : nullptr;
  double acc[kOhvTraits];
#pragma unroll
  for (int t = 0; t < kOhvTraits; ++t) acc[t] = 0.0;

  auto take = [&](int m, int bit) {
    if (h) h[m] = (uint8_t)bit;
    if (bit) {
#pragma unroll
      for (int t = 0; t < kOhvTraits; ++t)
        if (t < p.tcount) acc[t] += p.eff[(int64_t)(p.t0 + t) * p.effPitch + m];
    }
  };

  if (p.n_blocks == 0) {
    for (int m = lane; m < p.M; m += 64) {
      const uchar2 a = *reinterpret_cast<const uchar2 *>(g + 2 * (int64_t)m);
      const int a0 = a.x & 1, a1 = a.y & 1;
      take(m, p.eff[m] > 0.0 ? (a0 | a1) : (a0 & a1));
    }
  } else {
    for (int b = 0; b < p.n_blocks; ++b) {
      const int s = p.block_start[b], e = p.block_start[b + 1];
      double s0 = 0.0, s1 = 0.0;
      for (int m = s + lane; m < e; m += 64) {
        const uchar2 a = *reinterpret_cast<const uchar2 *>(g + 2 * (int64_t)m);
        const double w = p.effsum[m];
        s0 += (a.x & 1) ? w : 0.0;
        s1 += (a.y & 1) ? w : 0.0;
      }
      const int best = wave_sum(s1) > wave_sum(s0) ? 1 : 0;
      for (int m = s + lane; m < e; m += 64) take(m, g[2 * (int64_t)m + best] & 1);
    }
  }
  if (p.ohv) {
#pragma unroll
    for (int t = 0; t < kOhvTraits; ++t)
      if (t < p.tcount) {
        const double v = wave_sum(acc[t]);
        if (lane == 0) p.ohv[(int64_t)row * p.T + p.t0 + t] = (float)(2.0 * v);
      }
  }
}

}  // namespace bh
