// Multi-trait GEBV on the matrix cores, exactly.
//
//   GEBV[i, t] = sum_m dosage[i, m] * effects[m, t]           dosage in {0, 1, 2}
//
// is a (rows x M) x (M x T) product with tiny N, so its cost is reading the genotypes once -- if
// the effects are not re-read per row.  The f64 VALU path of row_kernels.h re-reads them for every
// row (or every 2-4 rows) and is VALU/L2-bound from T = 2 on.  Here the effects are turned, once
// per genetic map, into fixed point: eff * 2^s_t rounded to a 48-bit integer and cut into six
// balanced base-256 digits in [-128, 127]; the product then is six int8 GEMMs with int32
// accumulation (v_mfma_i32_32x32x32_i8), which is exact and order-independent, and
//       GEBV[i, t] = 2^-s_t * sum_d 256^d * S[i, t, d],   S = dosage (int8) x digit_d (int8).
// The only rounding is the 2^-48-relative quantisation of the effects -- far below f32 output
// precision -- so results agree with the f64 path to the last f32 bit except at exact ties.
//
// Work split: one wave = 32 rows (one MFMA row tile; two cost registers and halved the waves in
// flight for nothing: the kernel is latency-bound) x one slice of the marker axis; the
// (trait, digit) columns (T * 6 <= 64) are two 32-column tiles.  A wave step covers 64 markers:
// every lane loads 64 contiguous genotype bytes of its row (32 markers; lanes l and l+32 share a
// 128-byte line), packs them to 32 dosage bytes = the A operands of two MFMAs; the B operands come
// pre-packed in exactly the lane order they are consumed in.  Slices write int32 partial slabs;
// a second kernel adds the slabs, recombines the digits in int64 and scales.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bh {

constexpr int kGebvDigits = 6;
constexpr int kGebvCols = 64;  // two 32-column MFMA tiles; T * 6 <= 64  =>  T <= 10

using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;

struct GebvMfmaParams {
  const uint8_t *pop;  // (rows, pitch)
  const i32x4 *B;      // [chunk][ntile 2][mfma 2][lane 64] 16 int8 each: digits in consumption order
  int32_t *partial;    // [slices][rows_pad][64] int32
  int64_t pitch;
  int32_t rows, rows_pad;  // rows_pad = rows rounded up to 32
  int32_t n_chunks;        // ceil(M / 64)
  int32_t chunks_per_slice;
};

// 4 dosage bytes from two genotype dwords (2 markers each): bytes 0, 2 of w + (w >> 8)
__device__ __forceinline__ uint32_t dosage4(uint32_t w0, uint32_t w1) {
  const uint32_t s0 = w0 + (w0 >> 8), s1 = w1 + (w1 >> 8);
  return __builtin_amdgcn_perm(s1, s0, 0x06040200u);
}

__device__ __forceinline__ i32x4 dosage16(const uint4 a, const uint4 b) {
  i32x4 r;
  r.x = (int)dosage4(a.x, a.y);
  r.y = (int)dosage4(a.z, a.w);
  r.z = (int)dosage4(b.x, b.y);
  r.w = (int)dosage4(b.z, b.w);
  return r;
}

__global__ __launch_bounds__(256) void gebv_mfma_kernel(const GebvMfmaParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int row0 = (blockIdx.x * 4 + wave) * 32;  // 32-row tile
  if (row0 >= p.rows) return;
  const int slice = blockIdx.y;
  const int c_begin = slice * p.chunks_per_slice;
  const int c_end = min(p.n_chunks, c_begin + p.chunks_per_slice);

  // the row this lane feeds (clamped: rows past the end re-read the last row, never used)
  const uint8_t *pa = p.pop + (int64_t)min(row0 + r, p.rows - 1) * p.pitch + 64 * h;

  i32x16 acc[2];  // [column tile]
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[j][k] = 0;

  // two register sets, used in turn: the loads of chunk c+1 are in flight while chunk c is in
  // the matrix cores
  struct Chunk {
    uint4 ga[4];
    uint32_t keep[4];  // all ones where this lane's piece exists
    i32x4 b00, b01, b10, b11;  // [ntile][mfma]
  };
  auto load = [&](int c, Chunk &k) {
    const int64_t off = (int64_t)c * 128;  // byte offset of the chunk in a row; this lane: + 64 h
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // a 16-byte piece is either wholly inside the row or wholly beyond it (pitch % 16 == 0).  Every lane loads -- one
      // past the row's end its row's first piece, zeroed where it is used: a load under a lane condition is a branch,
      // and behind branches the compiler waits for every load in flight (gebv_bytes.h)
      const bool in = off + 64 * h + 16 * q + 16 <= p.pitch;
      k.keep[q] = in ? ~0u : 0u;
      k.ga[q] = *reinterpret_cast<const uint4 *>(pa + (in ? off + 16 * q : (int64_t)-64 * h));
    }
    const i32x4 *Bc = p.B + (int64_t)c * 256 + lane;
    k.b00 = Bc[0];
    k.b01 = Bc[64];
    k.b10 = Bc[128];
    k.b11 = Bc[192];
  };
  auto compute = [&](const Chunk &k) {
    // MFMA 0 sums markers 32h + 0..15 of the chunk, MFMA 1 markers 32h + 16..31 (A and B agree)
    i32x4 a0 = dosage16(k.ga[0], k.ga[1]), a1 = dosage16(k.ga[2], k.ga[3]);
    a0.x &= (int)k.keep[0], a0.y &= (int)k.keep[0], a0.z &= (int)k.keep[1], a0.w &= (int)k.keep[1];
    a1.x &= (int)k.keep[2], a1.y &= (int)k.keep[2], a1.z &= (int)k.keep[3], a1.w &= (int)k.keep[3];
    acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, k.b00, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, k.b10, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, k.b01, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, k.b11, acc[1], 0, 0, 0);
  };
  Chunk k0, k1;
  if (c_begin < c_end) load(c_begin, k0);
  for (int c = c_begin; c < c_end; c += 2) {
    if (c + 1 < c_end) load(c + 1, k1);
    compute(k0);
    if (c + 1 < c_end) {
      if (c + 2 < c_end) load(c + 2, k0);
      compute(k1);
    }
  }

  // C/D layout of the 32x32 MFMAs: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  int32_t *out = p.partial + ((int64_t)slice * p.rows_pad + row0) * kGebvCols;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int row = (k & 3) + 8 * (k >> 2) + 4 * h;
      out[(int64_t)row * kGebvCols + 32 * j + r] = acc[j][k];
    }
}

struct GebvFinishParams {
  const int32_t *partial;  // [slices][rows_pad][cols]
  const double *scale;     // (T) 2^-s_t
  float *gebv;             // (rows, Tstride)
  int32_t rows, rows_pad, slices, T, Tstride, t0;
  int32_t cols;            // slab row pitch in int32: 64 (gebv_mfma_kernel) or 32 * NT (gebv_bits_kernel)
};

// one wave per row, lane = column: the slabs are read in whole 256-byte rows; the six digit sums
// of a trait sit in six neighbouring lanes and are recombined with shuffles
__global__ __launch_bounds__(256) void gebv_finish_kernel(const GebvFinishParams p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  long long s = 0;
  if (lane < p.cols) {
    const int32_t *src = p.partial + (int64_t)row * p.cols + lane;
    const int64_t stride = (int64_t)p.rows_pad * p.cols;
    int sl = 0;
    for (; sl + 8 <= p.slices; sl += 8) {  // eight slabs in flight
      int32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(sl + u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sl < p.slices; ++sl) s += src[sl * stride];
  }
  long long total = 0;
#pragma unroll
  for (int d = 0; d < kGebvDigits; ++d) {
    const long long v = __shfl(s, (lane + d) & 63);
    total += v * (1LL << (8 * d));
  }
  const int t = lane / kGebvDigits;
  if (lane % kGebvDigits == 0 && t < p.T) p.gebv[(int64_t)row * p.Tstride + p.t0 + t] = (float)((double)total * p.scale[t]);
}

}  // namespace bh
