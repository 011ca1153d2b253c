// Device kernels for the BreedGym hot path on gfx950 (MI355X, CDNA4, wave64).
//
// One workgroup of WAVES wavefronts owns one output row (one cross / one gathered founder /
// one individual whose GEBV is wanted) and walks the marker axis:
//
//   phase A  (MODE_CROSS)  Threefry-2x32 per marker pair (j, j+half) for both gametes; the
//            Bernoulli(r) recombination sites leave the wave as ballots -> 64-bit site words
//            in LDS.  One Threefry call serves two markers (JAX's original random_bits
//            layout hashes the counter vector in two halves), and each threshold load is
//            shared by the two gametes.
//   phase B  inclusive prefix-XOR over the site bit strings (6 shift-xor steps inside a
//            64-bit word, ballot+popcount across words) -> crossover masks in LDS, stored
//            byte-interleaved so that phase C fetches both gametes' 8 mask bits with one
//            ds_read_u16.
//   phase C  16-byte coalesced reads of both parents (8 markers x 2 alleles per lane),
//            one v_perm_b32 per output dword picks the allele the masks point at, 16-byte
//            coalesced store of the offspring, and the GEBV dot product is accumulated in
//            f64 from the bytes already in registers (no re-read of the offspring).
//
// The parent gather of the reference (vec_env.py:89-90) is folded into the address of the
// phase-C loads: parents are never materialised.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "threefry.h"

namespace bh {

enum { MODE_CROSS = 0, MODE_GATHER = 1, MODE_GEBV = 2 };

struct RowParams {
  const uint8_t *src;    // cross: (E, n_pool, pitch) pool; gather: (n_pool, pitch); gebv: (rows, pitch)
  uint8_t *out;          // (rows, pitch) or null (gebv)
  const int32_t *idx;    // cross: (rows, 2); gather: (rows); gebv: null
  float *gebv;           // (rows, T) or null
  const uint32_t *thrA;  // [nwh*64] thresholds of markers [0, half), zero padded
  const uint32_t *thrB;  // [nwh*64] thresholds of markers [half, M), zero padded
  const double *eff;     // (tcount, effPitch) f64 marker effects of traits t0.., zero padded
  int64_t pitch;         // bytes per genotype row (multiple of 16)
  int64_t effPitch;      // doubles per trait (= 8 * nvec)
  int32_t M, half, nwh, nw, nvec;
  int32_t n_pool, n_out;  // rows per env in src / out (cross); n_pool = src rows (gather)
  int32_t n_rows;         // total output rows = grid size
  int32_t T, t0, tcount;  // GEBV row stride, first trait and trait count of this launch
  uint32_t key0, key1;    // the call's simulator key
  uint32_t nkeys;         // 2*B + 1 keys are split off it
  int32_t rng_shared;     // 1: key index from the env-local cross index
};

// Blocks b, b+8, b+16.. are dealt to the same XCD (round-robin dispatch).  Hand every XCD a
// contiguous run of rows so that the crosses of one env, which gather from the same parent
// pool, meet in one L2.  Bijective for any n; placement only changes speed.
__device__ __forceinline__ int xcd_row(int b, int n) {
  const int q = n >> 3, rem = n & 7, x = b & 7, s = b >> 3;
  return x * q + (x < rem ? x : rem) + s;
}

__device__ __forceinline__ int norm_index(int i, int n) {
  // JAX retrieval indexing: negative wraps once, then clamp
  if (i < 0) i += n;
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

__device__ __forceinline__ uint64_t prefix_xor64(uint64_t x) {
  x ^= x << 1;
  x ^= x << 2;
  x ^= x << 4;
  x ^= x << 8;
  x ^= x << 16;
  x ^= x << 32;
  return x;
}

// 64 site bits of markers [64w, 64w+64) out of the two half-strings.
// SA holds markers [0, half) from bit 0 of SA[0]; SB holds markers [half, M) from bit 0 of SB[0].
__device__ __forceinline__ uint64_t site_word(const uint64_t *SA, const uint64_t *SB, int w, int half) {
  const int base = w << 6;
  if (base + 64 <= half) return SA[w];
  if (base >= half) {
    const int off = base - half, k = off >> 6, s = off & 63;
    const uint64_t lo = SB[k];
    return s ? ((lo >> s) | (SB[k + 1] << (64 - s))) : lo;
  }
  return SA[w] | (SB[0] << (half - base));
}

template <int WAVES, int TT, int MODE>
__global__ __launch_bounds__(64 * WAVES) void row_kernel(const RowParams p) {
  extern __shared__ __align__(16) uint64_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c = xcd_row(blockIdx.x, p.n_rows);

  const uint8_t *src0 = nullptr, *src1 = nullptr;
  int e = 0, i = c;
  if constexpr (MODE == MODE_CROSS) {
    e = c / p.n_out;
    i = c - e * p.n_out;
    const int i0 = norm_index(p.idx[2 * (int64_t)c], p.n_pool);
    const int i1 = norm_index(p.idx[2 * (int64_t)c + 1], p.n_pool);
    src0 = p.src + ((int64_t)e * p.n_pool + i0) * p.pitch;
    src1 = p.src + ((int64_t)e * p.n_pool + i1) * p.pitch;
  } else if constexpr (MODE == MODE_GATHER) {
    src0 = p.src + (int64_t)norm_index(p.idx[c], p.n_pool) * p.pitch;
  } else {
    src0 = p.src + (int64_t)c * p.pitch;
  }

  // LDS carve-up (MODE_CROSS): site words of both gametes, then the interleaved masks
  uint64_t *SA0 = lds;
  uint64_t *SB0 = SA0 + p.nwh;
  uint64_t *SA1 = SB0 + p.nwh + 1;
  uint64_t *SB1 = SA1 + p.nwh;
  uint8_t *Z = reinterpret_cast<uint8_t *>(SB1 + p.nwh + 1);  // 16 * nw bytes, 16-B aligned
  double *red = reinterpret_cast<double *>(MODE == MODE_CROSS ? Z + 16 * (int64_t)p.nw : (uint8_t *)lds);

  if constexpr (MODE == MODE_CROSS) {
    // ---- gamete keys: keys = split(sim_key, nkeys); gamete (b,q) owns keys[1+2b+q];
    //      its recombination key is split(gamete_key, 2)[0].  Lanes 0..3 carry the 4 words.
    uint32_t rk00, rk01, rk10, rk11;
    {
      const uint32_t b = p.rng_shared ? (uint32_t)i : (uint32_t)c;
      const uint32_t q = (lane >> 1) & 1u, w = lane & 1u;
      const uint32_t kidx = 1u + 2u * b + q;
      const uint32_t gw = iota_word_even(p.key0, p.key1, 2u * p.nkeys, 2u * kidx + w);
      const uint32_t g0 = __shfl(gw, lane & ~1), g1 = __shfl(gw, lane | 1);
      const uint32_t rw = threefry2x32(g0, g1, w, w + 2u).x;
      rk00 = __builtin_amdgcn_readlane(rw, 0);
      rk01 = __builtin_amdgcn_readlane(rw, 1);
      rk10 = __builtin_amdgcn_readlane(rw, 2);
      rk11 = __builtin_amdgcn_readlane(rw, 3);
    }

    // ---- phase A: recombination sites
    for (int t = wv; t < p.nwh; t += WAVES) {
      const uint32_t j = (uint32_t)t * 64u + lane;
      const uint32_t tA = p.thrA[j], tB = p.thrB[j];
      uint32_t c1 = j + (uint32_t)p.half;
      if (c1 >= (uint32_t)p.M) c1 = 0u;  // the pad counter of an odd-length stream
      const u32x2 a = threefry2x32(rk00, rk01, j, c1);
      const u32x2 b = threefry2x32(rk10, rk11, j, c1);
      const uint64_t sA0 = __ballot((a.x >> 9) < tA);
      const uint64_t sB0 = __ballot((a.y >> 9) < tB);
      const uint64_t sA1 = __ballot((b.x >> 9) < tA);
      const uint64_t sB1 = __ballot((b.y >> 9) < tB);
      if (lane == 0) {
        SA0[t] = sA0;
        SB0[t] = sB0;
        SA1[t] = sA1;
        SB1[t] = sB1;
      }
    }
    if (tid == 0) {
      SB0[p.nwh] = 0;
      SB1[p.nwh] = 0;
    }
    __syncthreads();

    // ---- phase B: crossover masks = inclusive prefix-XOR of the site strings
    if (wv == 0) {
      uint32_t carry0 = 0, carry1 = 0;
      const uint64_t below = (1ull << lane) - 1ull;
      for (int w0 = 0; w0 < p.nw; w0 += 64) {
        const int w = w0 + lane;
        uint64_t x0 = 0, x1 = 0;
        if (w < p.nw) {
          x0 = prefix_xor64(site_word(SA0, SB0, w, p.half));
          x1 = prefix_xor64(site_word(SA1, SB1, w, p.half));
        }
        const uint64_t par0 = __ballot((x0 >> 63) != 0), par1 = __ballot((x1 >> 63) != 0);
        if ((__popcll(par0 & below) & 1u) ^ carry0) x0 = ~x0;
        if ((__popcll(par1 & below) & 1u) ^ carry1) x1 = ~x1;
        carry0 ^= __popcll(par0) & 1u;
        carry1 ^= __popcll(par1) & 1u;
        if (w < p.nw) {
          // bytes x0.b0 x1.b0 x0.b1 x1.b1 ... so that markers 8v..8v+7 are one u16 at Z+2v
          const uint32_t a0 = (uint32_t)x0, a1 = (uint32_t)(x0 >> 32);
          const uint32_t b0 = (uint32_t)x1, b1 = (uint32_t)(x1 >> 32);
          uint4 z;
          z.x = __builtin_amdgcn_perm(b0, a0, 0x05010400u);
          z.y = __builtin_amdgcn_perm(b0, a0, 0x07030602u);
          z.z = __builtin_amdgcn_perm(b1, a1, 0x05010400u);
          z.w = __builtin_amdgcn_perm(b1, a1, 0x07030602u);
          *reinterpret_cast<uint4 *>(Z + 16 * (int64_t)w) = z;
        }
      }
    }
    __syncthreads();
  }

  // ---- phase C: stream the row(s)
  double acc[TT > 0 ? TT : 1];
#pragma unroll
  for (int t = 0; t < (TT > 0 ? TT : 1); ++t) acc[t] = 0.0;

  uint8_t *dst = (MODE == MODE_GEBV) ? nullptr : p.out + (int64_t)c * p.pitch;
  const uint16_t *Z16 = reinterpret_cast<const uint16_t *>(Z);

  for (int v = tid; v < p.nvec; v += 64 * WAVES) {
    const uint4 a = *reinterpret_cast<const uint4 *>(src0 + 16 * (int64_t)v);
    uint4 o = a;
    if constexpr (MODE == MODE_CROSS) {
      const uint4 b = *reinterpret_cast<const uint4 *>(src1 + 16 * (int64_t)v);
      const uint32_t zz = Z16[v];
      // dword k holds markers 2k, 2k+1 as bytes [m0 a0, m0 a1, m1 a0, m1 a1]; output byte
      // (marker, p) = parent p's byte (marker, mask_p[marker]).  nibble = [g0(2k) g0(2k+1)
      // g1(2k) g1(2k+1)]; the multiply drops its bits onto bit 0 of selector bytes 0,2,1,3.
#define BH_SEL(k, aw, bw)                                                                   \
  __builtin_amdgcn_perm(                                                                    \
      bw, aw,                                                                               \
      ((((zz >> (2 * k)) & 3u) | (((zz >> (8 + 2 * k)) & 3u) << 2)) * 0x00208041u & 0x01010101u) | \
          0x06020400u)
      o.x = BH_SEL(0, a.x, b.x);
      o.y = BH_SEL(1, a.y, b.y);
      o.z = BH_SEL(2, a.z, b.z);
      o.w = BH_SEL(3, a.w, b.w);
#undef BH_SEL
    }
    if constexpr (MODE != MODE_GEBV) *reinterpret_cast<uint4 *>(dst + 16 * (int64_t)v) = o;

    if constexpr (TT > 0) {
      // dosage of markers (2k, 2k+1) = bytes 0 and 2 of  w + (w >> 8)   (alleles are 0/1)
      const uint32_t s0 = o.x + (o.x >> 8), s1 = o.y + (o.y >> 8);
      const uint32_t s2 = o.z + (o.z >> 8), s3 = o.w + (o.w >> 8);
      const double d0 = (double)(s0 & 0xffu), d1 = (double)((s0 >> 16) & 0xffu);
      const double d2 = (double)(s1 & 0xffu), d3 = (double)((s1 >> 16) & 0xffu);
      const double d4 = (double)(s2 & 0xffu), d5 = (double)((s2 >> 16) & 0xffu);
      const double d6 = (double)(s3 & 0xffu), d7 = (double)((s3 >> 16) & 0xffu);
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        if (t < p.tcount) {
          const double2 *ep = reinterpret_cast<const double2 *>(p.eff + (int64_t)t * p.effPitch + 8 * (int64_t)v);
          const double2 e0 = ep[0], e1 = ep[1], e2 = ep[2], e3 = ep[3];
          double s = acc[t];
          s = fma(d0, e0.x, s);
          s = fma(d1, e0.y, s);
          s = fma(d2, e1.x, s);
          s = fma(d3, e1.y, s);
          s = fma(d4, e2.x, s);
          s = fma(d5, e2.y, s);
          s = fma(d6, e3.x, s);
          s = fma(d7, e3.y, s);
          acc[t] = s;
        }
      }
    }
  }

  if constexpr (TT > 0) {
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      double s = acc[t];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      acc[t] = s;
    }
    if constexpr (WAVES > 1) {
      __syncthreads();  // everyone is done with Z before `red` (which may alias it) is written
      if (lane == 0)
        for (int t = 0; t < TT; ++t) red[wv * TT + t] = acc[t];
      __syncthreads();
      if (tid == 0)
        for (int t = 0; t < TT; ++t) {
          double s = 0.0;
          for (int w = 0; w < WAVES; ++w) s += red[w * TT + t];
          acc[t] = s;
        }
    }
    if (tid == 0) {
#pragma unroll
      for (int t = 0; t < TT; ++t)
        if (t < p.tcount) p.gebv[(int64_t)c * p.T + p.t0 + t] = (float)acc[t];
    }
  }
}

// reward[e] = max over the env's (individuals x traits) GEBV block (vec_env.py:97)
__global__ __launch_bounds__(256) void reward_max_kernel(const float *g, int64_t per_env, float *out) {
  __shared__ float part[4];
  const int e = blockIdx.x, tid = threadIdx.x;
  const float *ge = g + (int64_t)e * per_env;
  float m = -INFINITY;
  for (int64_t j = tid; j < per_env; j += 256) m = fmaxf(m, ge[j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((tid & 63) == 0) part[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) out[e] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}

// ---- top-k by bitonic sort in LDS --------------------------------------------------------
// key = (~orderable(score)) << 32 | index : ascending key order = descending score, ties to
// the lower index (jax.lax.top_k).  One block sorts one chunk of up to P (power of two)
// candidates of one row and emits its first kk (index, score) pairs.
__device__ __forceinline__ uint32_t orderable(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

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
  for (int size = 2; size <= p.P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int j = tid; j < (p.P >> 1); j += nt) {
        const int lo = ((j & ~(stride - 1)) << 1) | (j & (stride - 1));
        const int hi = lo + stride;
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

}  // namespace bh
