// GEBV of byte-layout rows on the matrix cores in ONE pass over the rows (`simulator.GEBV_model(population)`,
// breedgym/vector/vec_env.py:132-134, on the layout the reference's API uses: 2 M bytes per individual).
//
// The matrix-core pass of gebv_bits.h wants a lane to hold 16 markers of ITS row as one dword of bit planes; byte
// rows arrive the other way round if they are to be read at streaming rate -- a wave reads a contiguous KB of one
// row per instruction, lane l getting the 16 bytes (8 markers) of vector l.  Round 2 made that turn through HBM
// (pack_kernel wrote 1/8 of the rows back as bit planes, gebv_bits_kernel read them: 82 + 31 us at 2048 x 100k x 8).
// Here it goes through LDS: a wave packs the vectors of its 32 rows into bit-plane words (16 bytes -> 16 bits: the
// bit 0 of every byte), stores them as a 32 x 128-byte tile of planes (row stride 144: conflict-free 16-byte reads),
// and the MFMA phase reads its 64 bytes per lane from there instead of from memory -- the rest (six int8 digit
// GEMMs, digit operands staged once per workgroup, K cut into slices) is
// gebv_bits_kernel's, and so is the way out (gebv_tile_out).  Rows are read once, nothing else of their size moves: HBM-bound at 2 M bytes per individual.
//
// A chunk = 512 markers = 1 KB of a byte row = 128 bytes of planes.  The packing of chunk c+1 is spread over the
// four MFMA steps of chunk c (eight rows per step), its loads are issued two steps ahead (16 KB in flight per wave).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gebv_bits.h"

namespace bh {

struct GebvBytesParams {
  const uint8_t *rows;  // (n_rows, pitch) byte-layout rows
  const i32x4 *B;       // gebv_bits_kernel's digit operands: [mc][q 4][ntile 2][j 4][lane 64] 16 int8 each
  GebvOut out;
  int64_t pitch;
  int32_t n_rows;
  int32_t nvec;  // 16-byte vectors per row
  int32_t n_mc;
  int32_t row_groups, slices;
};

constexpr int kPlaneStride = 144;  // bytes from one row's planes to the next in LDS
static_assert(2 * 32 * kPlaneStride >= 32 * 64 * 4, "a wave's planes also hold its tile of digit sums (gebv_tile_out)");

// bit 0 of the 16 bytes of a genotype vector -> the vector's bit-plane word (bit i = byte i).  A dot product of four
// 0/1 bytes with (1, 2, 4, 8) IS their nibble: v_dot4_u32_u8, one full-rate instruction per dword, the second of a
// pair adding on with (16, 32, 64, 128)  (a multiply by a gathering constant is a quarter-rate v_mul_lo_u32).
__device__ __forceinline__ uint32_t plane_word(const uint4 v) {
  constexpr uint32_t kBit0 = 0x01010101u;
  const uint32_t lo = __builtin_amdgcn_udot4(v.y & kBit0, 0x80402010u, __builtin_amdgcn_udot4(v.x & kBit0, 0x08040201u, 0u, false), false);
  const uint32_t hi = __builtin_amdgcn_udot4(v.w & kBit0, 0x80402010u, __builtin_amdgcn_udot4(v.z & kBit0, 0x08040201u, 0u, false), false);
  return lo | (hi << 8);
}

template <int NT>
__global__ __launch_bounds__(256) void gebv_bytes_kernel(const GebvBytesParams p) {
  static_assert(NT == 1 || NT == 2, "one or two 32-column tiles");
  constexpr int QS = NT == 1 ? 2 : 1;      // MFMA steps per stage of digit operands (gebv_bits_kernel)
  constexpr int STAGE_V = NT * QS * 256;  // i32x4 per stage
  __shared__ i32x4 ldsB[2][STAGE_V];
  __shared__ __attribute__((aligned(16))) uint8_t planes[4][2][32 * kPlaneStride];
  int rg, slice;
  if (!slice_of_block(p.row_groups, p.slices, rg, slice)) return;  // uniform per workgroup
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int mc_begin = slice_begin(slice, p.slices, p.n_mc), mc_end = slice_begin(slice + 1, p.slices, p.n_mc);
  const int n_chunks = mc_end - mc_begin;  // > 0 by construction of the grid

  // the wave's 32 rows; a tile that reaches past the last row re-reads that row (gebv_tile_out adds nothing of it)
  const int row0 = (rg * 4 + wave) * 32;
  const int last = p.n_rows - 1;

  i32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[t][k] = 0;

  // ---- byte rows -> registers -> planes in LDS, eight rows at a time.  Every load is issued by every lane -- one
  // that has nothing to read (past the row's end, past the slice) reads the row's first vector and its word is
  // masked out when it is packed: a load under a lane condition becomes a branch, and behind the branches the
  // compiler waits for ALL loads in flight (vmcnt(0)) where it should leave the younger batch alone.
  uint4 ld[2][8];
  uint32_t keep[2];  // per batch in flight: 0xFFFF where this lane's vector exists
  auto issue = [&](int chunk, int b, int set) {  // rows 8 b .. 8 b + 7 of chunk `chunk` (wave-uniform)
    const int v = chunk * 64 + lane;
    const bool in = chunk < mc_end && v < p.nvec;
    keep[set] = in ? 0xFFFFu : 0u;
    const int64_t off = in ? 16 * (int64_t)v : 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = min(row0 + 8 * b + i, last);
      ld[set][i] = *reinterpret_cast<const uint4 *>(p.rows + (int64_t)row * p.pitch + off);
    }
  };
  auto pack = [&](int buf, int b, int set) {
    uint8_t *dst = &planes[wave][buf][0] + 2 * lane;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      *reinterpret_cast<uint16_t *>(dst + (8 * b + i) * kPlaneStride) = (uint16_t)(plane_word(ld[set][i]) & keep[set]);
  };

  // ---- digit operands: global -> registers -> LDS, two stages ahead (as in gebv_bits_kernel)
  const int n_stages = n_chunks * (4 / QS);
  i32x4 breg[NT * QS];
  auto load_b = [&](int k) {
    if (k < n_stages) {
      const i32x4 *src = p.B + ((int64_t)(mc_begin * 4 + k * QS) * 2) * 256;
#pragma unroll
      for (int u = 0; u < QS; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) breg[u * NT + t] = src[(u * 2 + t) * 256 + tid];
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NT * QS; ++u) ldsB[buf][u * 256 + tid] = breg[u];
  };
  uint4 g[4];  // this lane's 64 bytes of planes of the chunk in the MFMA phase
  auto step = [&](int buf, int u, int q) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      i32x4 b[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = ldsB[buf][(u * NT + t) * 256 + j * 64 + lane];
      const uint32_t x = j == 0 ? g[q].x : j == 1 ? g[q].y : j == 2 ? g[q].z : g[q].w;
      const i32x4 a = dosage16_from_bits(x);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b[t], acc[t], 0, 0, 0);
    }
  };

  // ---- prologue: chunk 0 packed outright, the first two batches of chunk 1 under way
  issue(mc_begin, 0, 0);
  issue(mc_begin, 1, 1);
  load_b(0);
  pack(0, 0, 0);
  issue(mc_begin, 2, 0);
  pack(0, 1, 1);
  issue(mc_begin, 3, 1);
  store_b(0);
  load_b(1);
  pack(0, 2, 0);
  issue(mc_begin + 1, 0, 0);
  pack(0, 3, 1);
  issue(mc_begin + 1, 1, 1);
  __syncthreads();  // stage 0 of the digit operands is in LDS

  int k = 0;  // stage of digit operands
  for (int c = 0; c < n_chunks; ++c) {
    const int cb = c & 1;
    {  // the planes this wave packed for chunk c: written by this wave alone, LDS operations of a wave are in order
      const uint8_t *src = &planes[wave][cb][0] + r * kPlaneStride + 64 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q) g[q] = *reinterpret_cast<const uint4 *>(src + 16 * q);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // batch q of chunk c + 1 -> planes (its loads were issued two steps ago); loads of the batch two steps on
      if (c + 1 < n_chunks) pack(cb ^ 1, q, q & 1);
      if (q < 2)
        issue(mc_begin + c + 1, q + 2, q & 1);
      else
        issue(mc_begin + c + 2, q - 2, q & 1);
      if (q % QS == 0) {
        store_b((k + 1) & 1);  // stage k+1's digits; that buffer was last read in stage k-1, before the barrier
        load_b(k + 2);
      }
      step(k & 1, q % QS, q);
      if (q % QS == QS - 1) {
        __syncthreads();
        ++k;
      }
    }
  }

  gebv_tile_out<NT>(acc, reinterpret_cast<int32_t *>(&planes[wave][0][0]), p.out, row0, slice, lane);
}

}  // namespace bh
