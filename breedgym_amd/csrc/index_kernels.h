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


// ---- round 5, second session: the same indices as a stream ------------------------------------------------------
// The kernel above walks a row two bytes per lane and trip (128 bytes per wave instruction), adds a marker's effects
// under a lane condition, and with blocks makes two dependent trips to memory per block: 2048 x 100k takes 1.0 ms
// without blocks, 2.0 ms with five per chromosome, 4.8-5.4 ms with eight traits -- 0.1-0.4 TB/s of genotype bytes
// (tools/time_ohv.py).  Here
//   * the optimal haploid comes out of a row as BITS: a 16-byte vector (eight markers) -> a bit-plane word as pack_kernel
//     makes it, any / all of the two alleles by two shifts, the choice between them (or between the haplotypes of a
//     block) by masks -- written as bytes where the caller wants them and, with both allele bits set to the haploid's,
//     as a bit-plane row;
//   * the value is then the matrix-core GEBV pass over those rows (gebv_bits.h): dosage = 2 x haploid, so the pass's
//     GEBV IS 2 sum_m OHP[m] e[m, t], exact in the fixed point, every trait in one pass;
//   * with blocks a workgroup first streams its rows once for the block values -- per 16-byte vector and haplotype one
//     f64 sum, added to the block's cell in LDS (ds_add_f64; over the wave first where its 512 markers lie in one block; a
//     vector that straddles a block edge adds marker by marker) --, decides every block (the first haplotype on ties,
//     like argmax), and streams the rows again for the bits.  Which block a marker lies in comes from a table (a uint16
//     per marker, ohp_blocks_kernel: a binary search per marker), read once for the ROWS rows a workgroup streams side
//     by side, like the effect sums.
constexpr int kOhpRows = 2;          // rows a workgroup streams side by side (the tables of a vector are read once for them)
constexpr int kOhpLdsBytes = 64 * 1024;  // block cells in LDS: 20 bytes per block and row of a workgroup

struct OhpParams {
  const uint8_t *pop;          // (n_rows, pitch) byte-layout rows
  const uint8_t *pos8;         // (nvec) bit i: the first trait's effect of marker 8 v + i is positive
  const double *effsum;        // (8 nvec) sum over traits, zero beyond M
  const uint16_t *blk;         // (8 nvec) block of every marker, 0xFFFF: none; null without blocks
  uint8_t *ohp;                // (n_rows, M) or null
  uint16_t *planes;            // (n_rows, nvp) the haploid in both allele bits, or null
  int64_t pitch;
  int32_t M, nvec, nvp, n_rows, n_blocks;
};

// block of every marker: the b with block_start[b] <= m < block_start[b + 1] (empty blocks are never met)
__global__ __launch_bounds__(256) void ohp_blocks_kernel(const int32_t *block_start, int n_blocks, int M, int n, uint16_t *blk) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= n) return;
  int lo = 0, hi = n_blocks + 1;  // first index whose start is > m
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (block_start[mid] <= m) lo = mid + 1; else hi = mid;
  }
  const int b = lo - 1;
  blk[m] = (uint16_t)((m < M && b >= 0 && b < n_blocks) ? b : 0xFFFF);
}

// bit 0 of the 16 bytes of a vector -> 16 bits (pack_kernel): even bits allele 0, odd bits allele 1 of eight markers
__device__ __forceinline__ uint32_t ohp_pack16(const uint4 x) {
  const uint32_t dw[4] = {x.x, x.y, x.z, x.w};
  uint32_t o = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) o |= (((dw[k] & 0x01010101u) * 0x01020408u) >> 24) << (4 * k);
  return o;
}
// 8 bits -> the even bits of 16
__device__ __forceinline__ uint32_t ohp_spread8(uint32_t x) {
  x = (x | (x << 4)) & 0x0F0Fu;
  x = (x | (x << 2)) & 0x3333u;
  return (x | (x << 1)) & 0x5555u;
}
// the haploid's eight bits (even positions of `e`) of vector v of a row -> where they are wanted
__device__ __forceinline__ void ohp_put(const OhpParams &p, int row, int v, uint32_t e) {
  if (p.planes) p.planes[(int64_t)row * p.nvp + v] = (uint16_t)(e | (e << 1));
  if (p.ohp) {
    uint8_t *h = p.ohp + (int64_t)row * p.M + 8 * (int64_t)v;
    const uint32_t a = e & 0x55u, b = (e >> 8) & 0x55u;
    const uint32_t lo = (a & 1u) | ((a & 4u) << 6) | ((a & 0x10u) << 12) | ((a & 0x40u) << 18);
    const uint32_t hi = (b & 1u) | ((b & 4u) << 6) | ((b & 0x10u) << 12) | ((b & 0x40u) << 18);
    if ((p.M & 7) == 0) {  // uniform: rows of the output start on eight bytes
      *reinterpret_cast<uint2 *>(h) = make_uint2(lo, hi);
    } else {
      const uint64_t w = ((uint64_t)hi << 32) | lo;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (8 * v + i < p.M) h[i] = (uint8_t)(w >> (8 * i));
    }
  }
}

// grid: ceil(n_rows / ROWS) workgroups of four waves, which share the ROWS rows' vectors out (a wave per row pair left the
// chip with four waves per CU at 2048 rows, each with 8 KB in flight: 2 TB/s); dynamic LDS: ROWS * 20 * n_blocks bytes
template <int ROWS>
__global__ __launch_bounds__(256) void ohp_stream_kernel(const OhpParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t ohp_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * ROWS;
  const uint8_t *g[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) g[r] = p.pop + (int64_t)min(row0 + r, p.n_rows - 1) * p.pitch;  // (a row past the end: the last one again, nothing of it is written)
  const int nb = p.n_blocks;
  double *cell = reinterpret_cast<double *>(ohp_lds);                                 // [row][block][haplotype]
  int32_t *best = reinterpret_cast<int32_t *>(ohp_lds + (size_t)ROWS * 16 * nb);     // [row][block]
  constexpr int U = 2;  // vectors per lane and trip, 64 apart

  if (nb > 0) {
    for (int i = threadIdx.x; i < ROWS * 2 * nb; i += 256) cell[i] = 0.0;
    __syncthreads();
    // ---- the block values: one pass over the rows
    for (int v0 = wave * 64 * U; v0 < p.nvec; v0 += 4 * 64 * U) {
      uint4 a[U][ROWS], bq[U];
      double w[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int v = min(v0 + 64 * u + lane, p.nvec - 1);  // every lane loads; a vector past the end is dropped below
        bq[u] = *reinterpret_cast<const uint4 *>(p.blk + 8 * (int64_t)v);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) a[u][r] = *reinterpret_cast<const uint4 *>(g[r] + 16 * (int64_t)v);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double2 e2 = *reinterpret_cast<const double2 *>(p.effsum + 8 * (int64_t)v + 2 * q);
          w[u][2 * q] = e2.x;
          w[u][2 * q + 1] = e2.y;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool in = v0 + 64 * u + lane < p.nvec;
        const uint32_t bw[4] = {bq[u].x, bq[u].y, bq[u].z, bq[u].w};
        const uint32_t b_first = bw[0] & 0xFFFFu, b_last = bw[3] >> 16;
        // The 512 markers of a wave's instruction mostly lie in ONE block: then the lanes' sums are added over the wave
        // and one lane adds to the cell -- 64 lanes adding to one LDS address are served one after the other (the
        // pass took 440 us of a 660 us call that way at 2048 x 100k, five blocks per chromosome).
        const uint32_t b_wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_first);
        const bool one_block = b_wave != 0xFFFFu && __all(!in || (b_first == b_wave && b_last == b_wave));
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          const uint32_t x = ohp_pack16(a[u][r]);
          if (one_block) {  // (uniform)
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              s0 += (x >> (2 * i)) & 1u ? w[u][i] : 0.0;
              s1 += (x >> (2 * i + 1)) & 1u ? w[u][i] : 0.0;
            }
            s0 = wave_sum(in ? s0 : 0.0);
            s1 = wave_sum(in ? s1 : 0.0);
            if (lane == 0) {
              atomicAdd(&cell[((int64_t)r * nb + b_wave) * 2], s0);
              atomicAdd(&cell[((int64_t)r * nb + b_wave) * 2 + 1], s1);
            }
            continue;
          }
          if (!in) continue;
          if (b_first == b_last) {  // the whole vector in one block (or in none)
            if (b_first == 0xFFFFu) continue;
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              s0 += (x >> (2 * i)) & 1u ? w[u][i] : 0.0;
              s1 += (x >> (2 * i + 1)) & 1u ? w[u][i] : 0.0;
            }
            atomicAdd(&cell[((int64_t)r * nb + b_first) * 2], s0);
            atomicAdd(&cell[((int64_t)r * nb + b_first) * 2 + 1], s1);
          } else {  // a block edge inside the vector: marker by marker
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const uint32_t b = (bw[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
              if (b == 0xFFFFu) continue;
              atomicAdd(&cell[((int64_t)r * nb + b) * 2], (x >> (2 * i)) & 1u ? w[u][i] : 0.0);
              atomicAdd(&cell[((int64_t)r * nb + b) * 2 + 1], (x >> (2 * i + 1)) & 1u ? w[u][i] : 0.0);
            }
          }
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ROWS * nb; i += 256) best[i] = cell[2 * i + 1] > cell[2 * i] ? 1 : 0;
    __syncthreads();
  }

  // ---- the haploid's bits: a pass over the rows
  constexpr int U2 = 8 / ROWS;  // 8 KB of genotype bytes per wave in flight
  for (int v0 = wave * 64 * U2; v0 < p.nvec; v0 += 4 * 64 * U2) {
    uint4 a[U2][ROWS], bq[U2];
    uint32_t pm[U2];
#pragma unroll
    for (int u = 0; u < U2; ++u) {
      const int v = min(v0 + 64 * u + lane, p.nvec - 1);
      if (nb > 0) bq[u] = *reinterpret_cast<const uint4 *>(p.blk + 8 * (int64_t)v);
      else pm[u] = p.pos8[v];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) a[u][r] = *reinterpret_cast<const uint4 *>(g[r] + 16 * (int64_t)v);
    }
#pragma unroll
    for (int u = 0; u < U2; ++u) {
      const int v = v0 + 64 * u + lane;
      if (v >= p.nvec) continue;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        if (row0 + r >= p.n_rows) continue;
        const uint32_t x = ohp_pack16(a[u][r]);
        uint32_t e;
        if (nb == 0) {
          const uint32_t any = (x | (x >> 1)) & 0x5555u, all = (x & (x >> 1)) & 0x5555u, pos = ohp_spread8(pm[u]);
          e = (any & pos) | (all & ~pos);
        } else {
          const uint32_t bw[4] = {bq[u].x, bq[u].y, bq[u].z, bq[u].w};
          const uint32_t b_first = bw[0] & 0xFFFFu, b_last = bw[3] >> 16;
          if (b_first == b_last) {  // the whole vector in one block (or in none: zeros)
            e = b_first == 0xFFFFu ? 0u : (x >> best[r * nb + b_first]) & 0x5555u;
          } else {
            e = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const uint32_t b = (bw[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
              if (b != 0xFFFFu) e |= ((x >> (2 * i + best[r * nb + b])) & 1u) << (2 * i);
            }
          }
        }
        ohp_put(p, row0 + r, v, e);
      }
    }
  }
  if (p.planes && wave == 0)  // the words behind the last marker, up to the pitch (fewer than 64: ctx.h, pw)
    for (int r = 0; r < ROWS; ++r)
      if (row0 + r < p.n_rows && p.nvec + lane < p.nvp) p.planes[(int64_t)(row0 + r) * p.nvp + p.nvec + lane] = 0;
}

}  // namespace bh
