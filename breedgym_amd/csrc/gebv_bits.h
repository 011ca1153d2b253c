// GEBV straight from bit-plane rows on the matrix cores, exactly.
//
//   GEBV[i, t] = sum_m dosage[i, m] * effects[m, t],   dosage = allele 0 + allele 1 of marker m
//
// with the effects in the same fixed point as gebv_mfma.h (six balanced base-256 digits of
// round(e * 2^s_t), exact int32 sums per digit, recombined in int64 by gebv_finish_kernel).  What is
// read per individual is its bit-plane row -- M/4 bytes instead of the 2M of the byte layout -- so the
// pass is no longer a question of genotype bytes: at 2048 x 100k x 8 traits the rows are 51 MB.  What
// it costs is (i) turning bits into int8 operands, (ii) the digit operand of every MFMA (1 KB per
// 32 x 32 x 32 product, 8x the genotype bytes it meets), (iii) the MFMAs.  So:
//
//   * a workgroup is four waves, each with its own 32-row tile(s), all walking the same slice of the
//     marker axis: the digit operands of a step are staged ONCE per workgroup in LDS (two buffers, the
//     global loads of step k+2 in flight while step k computes, one barrier per step) and read by all
//     four waves -- a quarter of the L2 traffic and none of its latency in the MFMA chain;
//   * a lane (r = row of the tile, h = half) owns 64 contiguous bytes of its row per 512-marker
//     macro-chunk (lanes r and r+32 share a 128-byte line; the four 16-byte loads are issued together, a
//     macro-chunk ahead); one dword of it = 16 markers = the A operand of one MFMA:
//         d = x - ((x >> 1) & 0x5555...)          16 two-bit dosages
//         operand dword q = (d >> 2q) & 0x03030303   (byte b = marker q + 4b: any order will do)
//     10 integer instructions per 16 markers and 32 rows;
//   * the K axis is cut into slices (rows alone never fill the chip: 2048 rows are 64 tiles).  A wave ends by
//     recombining the six digit sums of every (row, trait) of its tile into one 64-bit integer and adding it, with one
//     atomic, to the row group's accumulator; the LAST workgroup of a row group to arrive (a counter per row group)
//     scales the sums into the caller's GEBV and leaves accumulator and counter at zero for the next launch.  (Round 2
//     wrote an int32 slab per slice and added the slabs in a second kernel: 21 MB out and back and 5 us of launch at
//     2048 x 100k x 8.)
//
// Marker order inside an MFMA is whatever the lanes hold -- a sum over K does not care -- as long as
// the digit operand is packed to match: macro-chunk mc, step q (0..3), MFMA j (0..3), lane (c, h),
// byte i  <->  digit column c (+32 per column tile) of marker 512 mc + 256 h + 64 q + 16 j + (i >> 2) + 4 (i & 3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "gebv_mfma.h"

namespace bh {

constexpr int kPlaneStrideBits = 144;  // bytes from one row's 128 bytes of planes to the next in LDS (gebv_bytes.h: the same)
static_assert(32 * kPlaneStrideBits >= 33 * 32 * 4, "a wave's tile of planes also holds its tile of digit sums (gebv_tile_out)");

constexpr int kGsumTraits = 16;  // 64-bit sums per row in the accumulator (a pack has <= 10 traits; gebv_tile_out touches slots 0..10)

// where a pass leaves its result: every wave adds its tile to `gsum`, the last workgroup of a row group writes `gebv`
struct GebvOut {
  unsigned long long *gsum;  // [row / 32][trait 16][row % 32] fixed-point sums, zero between launches
  uint32_t *arrived;         // (row_groups) workgroups of the row group that have added their tiles, zero between launches
  const double *scale;       // (T) 2^-s_t of the pack's traits
  float *gebv;               // (n_rows, Tstride)
  int32_t n_rows, T, Tstride, t0;
};
__device__ __forceinline__ int64_t gsum_index(int row, int trait) {
  return ((int64_t)(row >> 5) * kGsumTraits + trait) * 32 + (row & 31);
}

// The wave's 32 x (32 NT) tile of digit sums (C/D layout of the 32x32 MFMAs: column = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) -> gsum.  Column 6 t + d holds digit d of trait t; trait 5 lies across
// the two column tiles and is added in two parts.  Sums are taken modulo 2^64 (the true total fits: s_t is chosen so).
// One atomic instruction adds 32 rows x 2 traits, the 32 rows of a trait side by side in memory: four cache lines
// (with a row's traits side by side it is 32 lines, and the atomics cost the 2048 x 100k x 8 pass 9 us of its 37).
// `tile`: 33 * 32 ints of LDS that belong to this wave.  Returns what the atomics returned, folded: a caller that
// has this value has its sums in memory (gebv_group_done).
template <int NT>
__device__ __forceinline__ unsigned long long gebv_tile_out(const i32x16 (&acc)[NT], int32_t *tile, const GebvOut &o, int row0,
                                                            int lane) {
  const int r = lane & 31, h = lane >> 5;
  unsigned long long seen = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int k = 0; k < 16; ++k) tile[((k & 3) + 8 * (k >> 2) + 4 * h) * 33 + r] = acc[t][k];
    // (a wave's LDS operations are served in order: the reads below see the writes above)
    const int first = t == 0 ? 0 : 5;  // first trait with a column in this tile
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int trait = first + 3 * h + u;
      unsigned long long s = 0;
#pragma unroll
      for (int d = 0; d < kGebvDigits; ++d) {
        const int col = 6 * trait + d - 32 * t;
        if (col >= 0 && col < 32) s += (unsigned long long)(long long)tile[r * 33 + col] << (8 * d);
      }
      // never under a condition: a lane with nothing to add adds 0 to its own slot (there are kGsumTraits slots in
      // every row of the accumulator, rows up to the last row group's end).  Under a lane condition every atomic
      // becomes a branch with a wait for its return behind it -- six round trips to memory in a row per tile.
      const bool mine = trait < o.T && row0 + r < o.n_rows;
      seen ^= atomicAdd(&o.gsum[gsum_index(row0 + r, trait)], mine ? s : 0ull);
    }
  }
  return seen;
}

// Every workgroup of row group `rg` calls this after its waves' gebv_tile_out; the one that arrives last writes the
// GEBVs of the group's rows and leaves sums and counter at zero.  Order: a wave's additions have RETURNED (`seen`
// is in its registers: they were carried out, wherever the memory system carries out atomics) before the barrier,
// the workgroup counts itself after it; so whoever reads the counter at slices - 1 comes after every addition, and
// reads the sums with atomics too (no cached copy).  No device-wide fence: on gfx950 that is a write-back and an
// invalidation of the XCD's whole L2 per workgroup -- the digit operands with it -- and doubled the pass's time; a
// release on the counter alone (the write-back, one wave per workgroup) still costs 6 to 24 us per launch
// (profiles/r03_ab_gebv_out.txt).
template <int GROUP_ROWS = 128, int THREADS = 256>
__device__ __forceinline__ void gebv_group_done(const GebvOut &o, unsigned long long seen, int rg, int slices, int tid, int *flag) {
  static_assert((GROUP_ROWS & (GROUP_ROWS - 1)) == 0, "rows of a row group: a power of two");
  asm volatile("; additions returned: %0" ::"v"(seen));
  __syncthreads();
  if (tid == 0) *flag = atomicAdd(&o.arrived[rg], 1u) == (uint32_t)(slices - 1);
  __syncthreads();
  if (!*flag) return;
  // all exchanges of a thread are issued before the first is used (one after the other they are a round trip to
  // memory each: 8 in a row held the last workgroups for 20 us)
  constexpr int ITEMS = (GROUP_ROWS * (kGebvCols / kGebvDigits) + THREADS - 1) / THREADS;
  long long v[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int i = tid + k * THREADS, row = rg * GROUP_ROWS + (i & (GROUP_ROWS - 1)), t = i / GROUP_ROWS;
    v[k] = 0;
    if (t < o.T && row < o.n_rows) v[k] = (long long)atomicExch(&o.gsum[gsum_index(row, t)], 0ull);
  }
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int i = tid + k * THREADS, row = rg * GROUP_ROWS + (i & (GROUP_ROWS - 1)), t = i / GROUP_ROWS;
    if (t < o.T && row < o.n_rows) o.gebv[(int64_t)row * o.Tstride + o.t0 + t] = (float)((double)v[k] * o.scale[t]);
  }
  if (tid == 0) atomicExch(&o.arrived[rg], 0u);
}

struct GebvBitsParams {
  const uint8_t *rows;  // (n_src, row_bytes) bit-plane rows
  const int32_t *idx;   // (n_rows) source row of output row i, or null (= i)
  const i32x4 *B;       // [mc][q 4][ntile 2][j 4][lane 64] 16 int8 each
  GebvOut out;
  int64_t row_bytes;    // 16 * ceil(M / 64)
  int32_t n_src, n_rows;
  int32_t n_mc;          // ceil(M / 512)
  int32_t row_groups, slices;  // the grid: 8 * row_groups * ceil(slices / 8) workgroups (slice_of_block)
};

// Which (row group, K slice) a workgroup takes.  Workgroups b, b + 8, ... share an XCD and with it an L2
// (MI355X_MICROARCH.md, workgroup dispatch): every XCD is handed whole K slices -- slice s goes to the blocks with
// b % 8 == s % 8, row groups innermost -- so that a slice's digit operands (32 KB per 512 markers; 6.4 MB for 100k
// markers, more than one L2 holds) come into that L2 once and are met there by every row group, instead of every
// XCD pulling the whole pack through the fabric once per row group.  Placement only changes speed.
__device__ __forceinline__ bool slice_of_block(int row_groups, int slices, int &rg, int &slice) {
  const int b = (int)blockIdx.x;
  if (slices < 8) {  // fewer slices than XCDs (many rows, short K): nothing to place, row groups over all XCDs
    rg = b % row_groups;
    slice = b / row_groups;
    return true;
  }
  const int i = b >> 3;
  rg = i % row_groups;
  slice = (i / row_groups) * 8 + (b & 7);
  return slice < slices;
}
// slice s of `slices` (<= n_mc) covers macro-chunks [slice_begin(s), slice_begin(s + 1)): lengths differ by one at most
__host__ __device__ __forceinline__ int slice_begin(int s, int slices, int n_mc) { return (int)((int64_t)s * n_mc / slices); }
// the grid that goes with it
inline unsigned slice_grid(int64_t row_groups, int64_t slices) {
  return (unsigned)(slices < 8 ? row_groups * slices : 8 * row_groups * ((slices + 7) / 8));
}

// 16 markers (one dword of a bit-plane row: bits 2k, 2k+1 = the alleles of marker k) -> 16 dosage bytes.
// The order of the bytes is free (the digit operand is packed to match), so no field is moved further than
// a shift of the whole dword: byte b of dword q is the dosage of marker q + 4 b.  10 instructions.
__device__ __forceinline__ i32x4 dosage16_from_bits(uint32_t x) {
  const uint32_t d = x - ((x >> 1) & 0x55555555u);  // 16 fields of 2 bits: 0, 1, 2
  i32x4 a;
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = (int)((d >> (2 * q)) & 0x03030303u);
  return a;
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose counter is a constant in the body
template <int... I, class Fn>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, Fn f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

template <int NT, int F>
__global__ __launch_bounds__(256) void gebv_bits_kernel(const GebvBitsParams p) {
  static_assert(NT == 1 || NT == 2, "one or two 32-column tiles");
  // a stage = the digit operands of QS steps (8 KB): with one column tile a step is only four MFMAs per tile,
  // too little between two barriers
  constexpr int QS = NT == 1 ? 2 : 1;
  constexpr int STAGE_V = NT * QS * 256;  // i32x4 per stage in LDS
  __shared__ i32x4 ldsB[2][STAGE_V];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  int rg, slice;
  if (!slice_of_block(p.row_groups, p.slices, rg, slice)) return;  // uniform per workgroup
  const int mc_begin = slice_begin(slice, p.slices, p.n_mc), mc_end = slice_begin(slice + 1, p.slices, p.n_mc);
  const int n_stages = (mc_end - mc_begin) * (4 / QS);  // > 0 by construction of the grid

  // every wave takes part in the staging and the barriers, also one whose tiles lie past the last row
  // (its lanes re-read the last row; gebv_tile_out adds nothing of it)
  //
  // How the rows are read.  The MFMA wants lane (r, h) to hold 64 bytes of row r per macro-chunk; read that way
  // (round 2) an instruction touches 32 rows and takes a quarter of a 128-byte line from each: four requests per
  // line at the L2.  Now an instruction takes WHOLE lines (8192 x 100k x 1: 61 -> 56 us, 2048 x 100k x 8: 31.6 -> 30.6):
  // lane l reads piece l & 7 of row 8 i + (l >> 3), i = 0..3, and the tile of 32 x 128 bytes is turned in LDS (row
  // stride 144: conflict-free 16-byte reads) -- written and read back by the same wave, no barrier.
  static_assert(F == 1, "one 32-row tile per wave");
  __shared__ __attribute__((aligned(16))) uint8_t turn[4][32 * kPlaneStrideBits];
  const int tile0 = rg * 4 + wave;
  const uint8_t *pa[4];  // the four rows this lane reads a piece of
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = min(tile0 * 32 + 8 * i + (lane >> 3), p.n_rows - 1);
    if (p.idx) {
      int s = p.idx[row];
      if (s < 0) s += p.n_src;
      row = s < 0 ? 0 : (s >= p.n_src ? p.n_src - 1 : s);
    }
    pa[i] = p.rows + (int64_t)row * p.row_bytes;  // the row's start: the piece offset is added per load (load_chunk)
  }

  i32x16 acc[F][NT];
#pragma unroll
  for (int f = 0; f < F; ++f)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[f][j][k] = 0;

  struct Chunk {
    uint4 g[F][4];
    uint32_t keep;  // all ones where this lane's pieces exist
  };
  auto load_chunk = [&](int mc, Chunk &c) {  // whole lines: c.g[0][i] = piece (lane & 7) of row 8 i + (lane >> 3)
    // a 16-byte piece is wholly inside the row or wholly beyond it (row_bytes % 16 == 0)
    // every lane loads -- one with nothing to read takes its row's first piece and zeroes it: a load under a lane
    // condition becomes a branch, and behind branches the compiler waits for every load in flight (gebv_bytes.h)
    // (the zeroing waits for turn_chunk: here it would wait for the loads)
    const bool in = mc < mc_end && (int64_t)mc * 128 + 16 * (lane & 7) + 16 <= p.row_bytes;
    // (the row's FIRST piece, whatever the lane: with the lane's own piece offset in the base a row of fewer than
    // eight pieces -- M <= 448 -- would be read past its end, and the last row past the end of the caller's buffer)
    const int64_t off = in ? (int64_t)mc * 128 + 16 * (lane & 7) : 0;
    c.keep = in ? ~0u : 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) c.g[0][i] = *reinterpret_cast<const uint4 *>(pa[i] + off);
  };
  auto turn_chunk = [&](Chunk &c) {  // ... -> c.g[0][q] = bytes [64 h + 16 q, +16) of row r: what the MFMA phase reads
    uint8_t *t = &turn[wave][0];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint4 v = c.g[0][i];
      *reinterpret_cast<uint4 *>(t + (8 * i + (lane >> 3)) * kPlaneStrideBits + 16 * (lane & 7)) =
          make_uint4(v.x & c.keep, v.y & c.keep, v.z & c.keep, v.w & c.keep);
    }
    // (a wave's LDS operations are served in order: its reads below see its writes above)
#pragma unroll
    for (int q = 0; q < 4; ++q) c.g[0][q] = *reinterpret_cast<const uint4 *>(t + r * kPlaneStrideBits + 64 * h + 16 * q);
  };
  // digit operands of stage k of this slice -> registers (NT * QS 16-byte pieces per thread), kRing stages ahead of
  // the stage that reads them from LDS: a stage is eight MFMAs long, an L2 hit under load ten times that -- asked
  // for two stages ahead (round 2) the pass ran at the pace of its operand loads.  Stage s lives in register set
  // s % kRing; the sets are named by constants (the loop body below is kRing stages or a multiple).
  constexpr int kRing = 4;
  i32x4 breg[kRing][NT * QS];
  auto load_b = [&](int k, auto set) {
    const int kk = min(k, n_stages - 1);  // past the slice: its last stage again, never used (no branch: load_chunk)
    const i32x4 *src = p.B + ((int64_t)(mc_begin * 4 + kk * QS) * 2) * 256;
#pragma unroll
    for (int u = 0; u < QS; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) breg[decltype(set)::value][u * NT + t] = src[(u * 2 + t) * 256 + tid];
  };
  auto store_b = [&](int buf, auto set) {
#pragma unroll
    for (int u = 0; u < NT * QS; ++u) ldsB[buf][u * 256 + tid] = breg[decltype(set)::value][u];
  };
  auto step = [&](int buf, int u, const Chunk &c, int q) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      i32x4 b[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = ldsB[buf][(u * NT + t) * 256 + j * 64 + lane];
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const uint4 g = c.g[f][q];
        const uint32_t x = j == 0 ? g.x : j == 1 ? g.y : j == 2 ? g.z : g.w;
        const i32x4 a = dosage16_from_bits(x);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[f][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b[t], acc[f][t], 0, 0, 0);
      }
    }
  };
  constexpr int kChunkStages = 4 / QS;
  static_assert((2 * kChunkStages) % kRing == 0, "the loop body (two macro-chunks) is whole turns of the ring");
  auto chunk_stages = [&](int &k, Chunk &c, auto first) {  // `first`: this chunk's first stage, modulo kRing
    turn_chunk(c);
    static_for<kChunkStages>([&](auto qq) {
      const std::integral_constant<int, (decltype(first)::value + decltype(qq)::value + 1) % kRing> set;  // of stage k + 1
      store_b((k + 1) & 1, set);  // its LDS buffer was last read in stage k-1, before the barrier
      load_b(k + 1 + kRing, set);
#pragma unroll
      for (int u = 0; u < QS; ++u) step(k & 1, u, c, decltype(qq)::value * QS + u);
      __syncthreads();
      ++k;
    });
  };

  Chunk c0, c1;
  load_chunk(mc_begin, c0);
  static_for<kRing>([&](auto s) { load_b(decltype(s)::value, s); });
  store_b(0, std::integral_constant<int, 0>{});
  load_b(kRing, std::integral_constant<int, 0>{});
  __syncthreads();
  int k = 0;
  for (int mc = mc_begin; mc < mc_end; mc += 2) {
    load_chunk(mc + 1, c1);
    chunk_stages(k, c0, std::integral_constant<int, 0>{});
    if (mc + 1 < mc_end) {
      load_chunk(mc + 2, c0);
      chunk_stages(k, c1, std::integral_constant<int, kChunkStages % kRing>{});
    }
  }

  __shared__ int last_flag;
  const unsigned long long seen = gebv_tile_out<NT>(acc[0], reinterpret_cast<int32_t *>(&turn[wave][0]), p.out, tile0 * 32, lane);
  gebv_group_done(p.out, seen, rg, p.slices, tid, &last_flag);
}

}  // namespace bh
