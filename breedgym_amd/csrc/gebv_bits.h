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

constexpr int kGsumTraits = 10;  // 64-bit cells per row in the accumulators: the traits of a pack

// Where a pass leaves its result.  The marker axis is cut into `slices`, every workgroup holds the sums of its slice
// for its rows, and they have to meet.  Round 3 added them with returned atomics to ONE accumulator per (row, trait)
// and counted workgroups per row group on the side; the last one to arrive read the sums back.  But same-address
// atomics are served one after the other at the memory side (~170 ns each), all slices of a row group finish together,
// and so every workgroup sat out a chain of `slices` additions (49 x 170 ns), then a barrier, a counter, and the last
// one ten exchanges per thread: at 2048 x 100k x 8 the way out was HALF the pass (21 of 42 us with one round of
// workgroups; profiles/r04_ab_gebv.txt).  Now
//   * the arrival count rides in the sum: a cell is sum + 2^56 * arrivals (the fixed point leaves |sum| < 2^55, see
//     bh_create), so the returned value of a workgroup's own addition says whether it was the last -- no counter, no
//     barrier, no second look: whoever completes a cell carries it on and leaves it zero;
//   * two levels: the slices are taken in groups of ~sqrt(slices); a slice adds to its group's cell (`acc1`), the last
//     of a group adds the group's sum to the final cell (`acc2`), the last there writes the GEBV.  The longest chain
//     of same-address additions is 2 sqrt(slices) instead of slices.
// Every cell is zero whenever no launch is in flight.
constexpr int kArrivalShift = 56;
struct GebvOut {
  unsigned long long *acc1;  // [slice group][row / 32][trait 16][row % 32]
  unsigned long long *acc2;  // [row / 32][trait 16][row % 32]
  const double *scale;       // (T) 2^-s_t of the pack's traits
  float *gebv;               // (n_rows, Tstride)
  int64_t group_stride;      // cells per slice group in acc1
  int32_t n_rows, T, Tstride, t0;
  int32_t slices, group_size, n_groups;  // slice s belongs to group s / group_size
};
__device__ __forceinline__ int64_t gsum_index(int row, int trait) {
  return ((int64_t)(row >> 5) * kGsumTraits + trait) * 32 + (row & 31);
}
// arrivals before this addition, from what the addition returned (the sum field is signed: bias it first)
__device__ __forceinline__ uint32_t arrivals_of(unsigned long long old) {
  return (uint32_t)((old + (1ull << (kArrivalShift - 1))) >> kArrivalShift);
}

// The wave's F tiles of 32 x (32 NT) digit sums (C/D layout of the 32x32 MFMAs: column = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) -> the accumulators, and on to the GEBV where this was the last
// slice.  Column 6 t + d holds digit d of trait t.  A tile is turned through LDS (32 rows x 64 columns of int32, the
// column XOR-ed with the row so that neither the writes -- a register of 32 columns of one row -- nor the reads -- one
// column of 32 rows -- meet a bank twice), lane (r, h) picks up the six digits of trait 2 i + h of row r and adds them
// as ONE 64-bit number: ceil(T / 2) atomic instructions per tile, every active lane a real (row, trait) -- the memory
// side adds ~1.3 TB/s of atomic bytes whatever the addresses (MI355X_MICROARCH.md, global atomics), so what is added
// is what the way out costs; round 3's form added 12 cells per row for 8 traits (a cell per three-trait slot of each
// half-tile, padding included, trait 5 in two parts).  One instruction covers 32 rows of two traits, the rows of a
// trait side by side in memory (four cache lines).  Sums are taken modulo 2^64 (the true total fits).
// `tile`: 8 KB of LDS that belong to this wave.
template <int NT, int F>
__device__ __forceinline__ void gebv_tiles_out(const i32x16 (&acc)[F][NT], int32_t *tile, const GebvOut &o, int row0, int slice, int lane) {
  constexpr int kInstr = NT == 2 ? 5 : 3;  // traits of a pack: <= 10 with two column tiles, <= 5 with one
  const int r = lane & 31, h = lane >> 5;
  const int grp = slice / o.group_size;
  const int grp_n = min(o.group_size, o.slices - grp * o.group_size);  // slices of this group
  unsigned long long *a1 = o.acc1 + (int64_t)grp * o.group_stride;
  constexpr unsigned long long kOne = 1ull << kArrivalShift;
  unsigned long long mine[F][kInstr], old[F][kInstr];
#pragma unroll
  for (int f = 0; f < F; ++f) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int row = (k & 3) + 8 * (k >> 2) + 4 * h;
        tile[row * 64 + ((r + 32 * t) ^ row)] = acc[f][t][k];
      }
    // (a wave's LDS operations are served in order: the reads below see the writes above, the next tile's writes
    // come behind these reads)
#pragma unroll
    for (int i = 0; i < kInstr; ++i) {
      const int trait = 2 * i + h;
      unsigned long long s = 0;
#pragma unroll
      for (int d = 0; d < kGebvDigits; ++d) {
        const int col = min(6 * trait + d, 32 * NT - 1);  // (a trait beyond the pack: any column, nothing is added)
        s += (unsigned long long)(long long)tile[r * 64 + (col ^ r)] << (8 * d);
      }
      // Every lane adds, none under a condition (a lane condition turns an atomic into a branch with a wait for its
      // return behind it): a trait beyond T or a row beyond the last adds to a cell of its own that exists for it
      // (kGsumTraits traits, whole tiles of rows), like everybody else, and its sum goes nowhere.  Instructions whose
      // two traits are both beyond T are left out as a whole (uniform: T is the launch's).
      mine[f][i] = s;
      if (2 * i < o.T) old[f][i] = atomicAdd(&a1[gsum_index(row0 + 32 * f + r, trait)], s + kOne);
    }
  }
  // ... and only now is anything looked at: all the additions of the wave are in flight together.  Whoever completed
  // a cell of its group carries the group's sum on; that is rare (one slice of a group), so it is decided for the wave
  // as a whole, and a wave that carries anything on does so for all its cells at once, in flight together again -- a
  // lane whose cell is not complete adds 0 to the final cell, which changes neither its sum nor its count (one after
  // the other, under a lane condition each, these second additions were ten trips to memory in a row for the last
  // workgroups: 15 us of a 34 us pass)
  bool last1[F][kInstr];
  bool any = false;
#pragma unroll
  for (int f = 0; f < F; ++f)
#pragma unroll
    for (int i = 0; i < kInstr; ++i) {
      last1[f][i] = 2 * i < o.T && arrivals_of(old[f][i]) + 1u == (uint32_t)grp_n;
      any |= last1[f][i];
    }
  if (!__any(any)) return;
  unsigned long long gsum[F][kInstr], old2[F][kInstr];
#pragma unroll
  for (int f = 0; f < F; ++f)
#pragma unroll
    for (int i = 0; i < kInstr; ++i) {
      if (2 * i >= o.T) continue;  // uniform
      const int64_t cell = gsum_index(row0 + 32 * f + r, 2 * i + h);
      gsum[f][i] = old[f][i] + mine[f][i] - (unsigned long long)arrivals_of(old[f][i]) * kOne;  // the group's sum
      // nobody else comes to this cell in this launch; zeroed where the additions are made (device scope), not by a plain
      // store that would sit in this XCD's L2 until the kernel ends
      if (last1[f][i]) __hip_atomic_store(&a1[cell], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old2[f][i] = atomicAdd(&o.acc2[cell], last1[f][i] ? gsum[f][i] + kOne : 0ull);
    }
#pragma unroll
  for (int f = 0; f < F; ++f)
#pragma unroll
    for (int i = 0; i < kInstr; ++i) {
      if (2 * i >= o.T) continue;
      const int trait = 2 * i + h, row = row0 + 32 * f + r;
      const uint32_t before2 = arrivals_of(old2[f][i]);
      if (!last1[f][i] || before2 + 1u != (uint32_t)o.n_groups) continue;
      const long long total = (long long)(old2[f][i] + gsum[f][i] - (unsigned long long)before2 * kOne);
      // (stragglers of carrying waves may still add 0 to this cell: exactly 0, so the order does not matter)
      __hip_atomic_store(&o.acc2[gsum_index(row, trait)], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (trait < o.T && row < o.n_rows) o.gebv[(int64_t)row * o.Tstride + o.t0 + trait] = (float)((double)total * o.scale[trait]);
    }
}
// one tile
template <int NT>
__device__ __forceinline__ void gebv_tile_out(const i32x16 (&acc)[NT], int32_t *tile, const GebvOut &o, int row0, int slice, int lane) {
  gebv_tiles_out<NT, 1>(reinterpret_cast<const i32x16(&)[1][NT]>(acc), tile, o, row0, slice, lane);
}

struct GebvBitsParams {
  const uint8_t *rows;  // (n_src, row_bytes) bit-plane rows
  const int32_t *idx;   // (n_rows) source row of output row i, or null (= i)
  const i32x4 *B;       // [mc][q 4][ntile 2][j 4][lane 64] 16 int8 each
  GebvOut out;
  int64_t row_bytes;    // bh_packed_pitch(): 16 * ceil(M / 64), whole 128-byte lines beyond 4 KB
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

// ---- the pass (round 4) ----------------------------------------------------------------------------------------
// A workgroup is EIGHT waves, each with TWO 32-row tiles (512 rows per workgroup), all walking the same slice of the
// marker axis.  What bounded round 3's form (four waves, one tile each, two workgroups per CU) was the LDS: per 128
// markers a workgroup read its 8 KB of digit operands four times (once per wave), wrote them once out of registers,
// and turned 4 KB of rows through it -- 48 KB per stage, 0.77 of the LDS's 128 bytes per clock with two workgroups on
// the CU; the MFMAs were waiting for it.  Now
//   * an operand read serves two MFMAs (the wave's two row tiles): half the LDS reads per product;
//   * the digit operands come by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write), a superstage
//     (256 markers: 8 NT KB) at a time into a ring of three, every wave fetching its NT KB of it two superstages
//     ahead; ONE barrier per superstage (32 NT MFMAs per wave between two of them);
//   * the rows come by LDS-DMA too, whole 128-byte lines (lane l: piece (l & 7) ^ (l >> 3) of row 8 i + (l >> 3), so
//     that the MFMA phase's 16-byte reads -- lane (r, h) takes pieces 4 h .. 4 h + 3 of row r -- meet no bank twice),
//     into a slot of the wave's own, a macro-chunk ahead: the slot is read out into registers at the start of a
//     macro-chunk and re-armed at once.  No barrier, no turn through registers.
// Waits are counted by hand (vmcnt counts the DMAs in order; every wave issues the same DMAs in every superstage, none
// under a condition: past the slice's end they fetch the next slice's operands or the pack's trailing zero
// macro-chunk, and a row's start again) and the LDS reads of DMA-written data are inline asm, so that the compiler,
// which would wait for the NEWEST DMA before any LDS read, stays out of it (packed_dma.h has the same arrangement).
// Digits beyond marker M are zero in the pack, so bytes beyond a row's end may be anything: only their address is
// kept inside the row.  (Rows of more than 4 KB start on 128-byte lines -- bh_packed_pitch -- so a macro-chunk of a row
// IS a line.  At the old pitch, 25,008 bytes for 100,000 markers, every piece of 128 bytes straddled two and the pass
// over 8192 rows took 52 us instead of 40 with its rows in the Infinity Cache: profiles/r05_ab_pitch.txt.  Measured
// there as well, and no help inside the step: two macro-chunks of rows in flight instead of one, and pulling the
// rows' lines into the L2 ahead of the DMAs with a 4-byte LDS-DMA per row.)
constexpr int kBitsF = 2;               // 32-row tiles per wave
constexpr int bits_rows(int waves) { return waves * kBitsF * 32; }  // rows per workgroup = rows per row group
constexpr int kBitsRing = 3;            // superstages of digit operands in LDS
constexpr int kBitsRowSlot = kBitsF * 4096;  // bytes of LDS per wave: its two tiles' 32 x 128 bytes of one macro-chunk
// row slots per wave: with one column tile (a pass bound by its rows: 8192 x 100k x 1 reads 205 MB) there is room for
// two, and the rows are asked for two macro-chunks ahead
template <int NT, int WAVES>
constexpr int bits_row_slots() {
  return NT == 1 && WAVES == 8 ? 2 : 1;
}
template <int NT, int WAVES>
constexpr int gebv_bits_lds() {
  return WAVES * kBitsRowSlot * bits_row_slots<NT, WAVES>() + kBitsRing * 8192 * NT;
}
static_assert(gebv_bits_lds<1, 8>() <= 160 * 1024 && gebv_bits_lds<2, 8>() <= 160 * 1024, "one workgroup's LDS");
static_assert(kBitsRowSlot >= 32 * 64 * 4, "a wave's row slot also holds a tile of digit sums (gebv_tiles_out)");

// s_waitcnt vmcnt(n) alone (gfx9 encoding; packed_dma.h: vmcnt_only)
__host__ __device__ constexpr int bits_vmcnt(int n) { return (n & 0xF) | ((n >> 4) << 14) | 0x0F70; }

// One LDS-DMA instruction: lane l's 16 bytes at `g` -> LDS byte lds_base + 16 l (lds_base wave-uniform).  Inline asm on
// purpose: told about the DMA (the builtin), the compiler makes every later LDS read wait for the NEWEST DMA in flight
// (vmcnt(0)); not told, it counts its own ds_reads with lgkmcnt and leaves vmcnt to the hand-placed waits below.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_dma16(const void *g, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_base) : "memory", "m0");
}
#pragma clang diagnostic pop
// nothing moves across this line in the compiler's schedule of memory operations (emits nothing)
#define BH_COMPILER_FENCE() asm volatile("" ::: "memory")

template <int NT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gebv_bits_kernel(const GebvBitsParams p) {
  static_assert(NT == 1 || NT == 2, "one or two 32-column tiles");
  static_assert(WAVES == 4 || WAVES == 8, "the waves share out the 8 NT KB of a superstage of digit operands");
  constexpr int NB = 8 * NT / WAVES;  // DMA instructions of digit operands per wave and superstage
  constexpr int F = kBitsF, RS = kBitsRing, SLOTS = bits_row_slots<NT, WAVES>();
  extern __shared__ __attribute__((aligned(1024))) uint8_t bits_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  int rg, slice;
  if (!slice_of_block(p.row_groups, p.slices, rg, slice)) return;  // uniform per workgroup
  // A slice is a run of SUPERSTAGES (half macro-chunks: with 196 macro-chunks over 64 slices, whole ones come out 3 or 4
  // long and the launch lasts as long as the 4; halves come out 6 or 7).  The stream below still moves whole macro-chunks
  // -- rows come in 128-byte lines, and the counted waits want every macro-chunk alike -- and a superstage outside
  // [ss_begin, ss_end) has its operands fetched like any other and its products left out (uniform per workgroup).
  const int ss_begin = slice_begin(slice, p.slices, 2 * p.n_mc), ss_end = slice_begin(slice + 1, p.slices, 2 * p.n_mc);
  const int mc_begin = ss_begin >> 1, mc_end = (ss_end + 1) >> 1;
  const int n_mc = mc_end - mc_begin;  // macro-chunks this slice touches, > 0 by construction of the grid (slices <= n_mc)
  // The row groups of a slice run side by side on one XCD and want the same digit operands: started on the same
  // macro-chunk they ask for every line of them at the same moment, and each request goes out to the fabric (PMC: 130 MB
  // per launch at 2048 x 100k x 8 where rows and operands are 58).  A sum does not care for order: row group rg walks
  // its slice from macro-chunk rg on, round the end -- one workgroup brings a line into the L2, the others find it there.
  const int rot = rg % n_mc;
  auto mc_of = [&](int m) { return mc_begin + (m + rot) % n_mc; };  // the m-th macro-chunk of this workgroup's walk (any m >= 0)

  uint8_t *rowslot = bits_lds + wave * (kBitsRowSlot * SLOTS);    // wave-uniform; slot s at + s kBitsRowSlot
  uint8_t *ring = bits_lds + WAVES * kBitsRowSlot * SLOTS;  // workgroup-uniform
  const uint32_t rowslot_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)rowslot;
  const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)ring;

  // every wave takes part in the staging and the barriers, also one whose tiles lie past the last row
  // (its lanes re-read the last row; gebv_tile_out's sums of it go nowhere)
  const int row0 = rg * bits_rows(WAVES) + wave * (32 * F);
  const uint8_t *rp[F][4];  // the rows this lane fetches a piece of: tile f, rows 8 i + (lane >> 3)
  {
    int row[F * 4];
#pragma unroll
    for (int u = 0; u < F * 4; ++u) row[u] = min(row0 + 8 * u + (lane >> 3), p.n_rows - 1);
    if (p.idx) {  // all eight indices asked for at once (one after the other they are eight trips to memory)
      int s[F * 4];
#pragma unroll
      for (int u = 0; u < F * 4; ++u) s[u] = p.idx[row[u]];
#pragma unroll
      for (int u = 0; u < F * 4; ++u) {
        const int v = s[u] < 0 ? s[u] + p.n_src : s[u];
        row[u] = v < 0 ? 0 : (v >= p.n_src ? p.n_src - 1 : v);
      }
    }
#pragma unroll
    for (int u = 0; u < F * 4; ++u) rp[u >> 2][u & 3] = p.rows + (int64_t)row[u] * p.row_bytes;
  }
  const int piece = (lane & 7) ^ (lane >> 3);  // of the macro-chunk's 128-byte line: LDS unit (lane & 7) of its row
  auto dma_rows = [&](int mc, int rs) {  // macro-chunk mc of the wave's 64 rows -> its row slot rs (4 F instructions)
    const int64_t want = (int64_t)mc * 128 + 16 * piece;
    const int64_t off = want + 16 <= p.row_bytes ? want : 0;  // a piece beyond the row: the row's first one (never used: zero digits)
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int i = 0; i < 4; ++i) lds_dma16(rp[f][i] + off, rowslot_lds + rs * kBitsRowSlot + f * 4096 + i * 1024);
  };
  auto dma_b = [&](int k, int ring_slot) {  // this wave's share of superstage k of the slice -> that ring slot (NB instructions)
    // [mc][q 4][ntile 2][j 4][lane 64] x 16 bytes: a superstage = q 2 kk, 2 kk + 1 of one macro-chunk = 16 KB in a row
    const i32x4 *src = p.B + ((int64_t)mc_of(k >> 1) * 4 + 2 * (k & 1)) * 2 * 256;
    const uint32_t dst = ring_lds + ring_slot * (8192 * NT);
    if constexpr (NT == 2) {
#pragma unroll
      for (int u = 0; u < NB; ++u)  // the slot mirrors memory: [q' 2][tile 2][j 4] KB
        lds_dma16(src + (NB * wave + u) * 64 + lane, dst + (NB * wave + u) * 1024);
    } else {  // tile 0 only: the slot is [q' 2][j 4] KB
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int unit = NB * wave + u, qq = unit >> 2, j = unit & 3;
        lds_dma16(src + (qq * 2) * 256 + j * 64 + lane, dst + unit * 1024);
      }
    }
  };

  i32x16 acc[F][NT];
#pragma unroll
  for (int f = 0; f < F; ++f)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[f][t][k] = 0;

  // where lane (r, h) finds piece 4 h + q of row r in a tile of the row slot; where it finds its digit operands
  const uint4 *gp[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    gp[q] = reinterpret_cast<const uint4 *>(rowslot + (r >> 3) * 1024 + ((r & 7) * 8 + ((4 * h + q) ^ (r & 7))) * 16);
  const i32x4 *ringp = reinterpret_cast<const i32x4 *>(ring) + lane;
  auto read_rows = [&](uint4 (&g)[F][4], int rs) {
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) g[f][q] = gp[q][rs * (kBitsRowSlot / 16) + f * 256];
  };
  auto read_b = [&](int ring_slot, int qq, i32x4 (&b)[NT][4]) {
    const i32x4 *src = ringp + ring_slot * (512 * NT) + qq * (256 * NT);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) b[t][j] = src[(t * 4 + j) * 64];
  };
  auto products = [&](const uint4 (&g)[F][4], auto q_c, const i32x4 (&b)[NT][4]) {  // 128 markers x the wave's 64 rows
    constexpr int q = decltype(q_c)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const uint4 gq = g[f][q];
        const uint32_t x = j == 0 ? gq.x : j == 1 ? gq.y : j == 2 ? gq.z : gq.w;
        const i32x4 a = dosage16_from_bits(x);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[f][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b[t][j], acc[f][t], 0, 0, 0);
      }
  };

  // ---- the stream.  DMAs in flight per wave, in issue order (vmcnt counts them in that order); k = 2 m is the even
  // superstage of macro-chunk m, A = SLOTS - 1 further macro-chunks of rows are asked for ahead:
  //   ... B(k) [NB]  rows(m + 1 + A) [4 F]  B(k + 1) [NB]  B(k + 2) [NB]  rows(m + 2 + A) [4 F]  B(k + 3) ...
  constexpr int A = SLOTS - 1, RD = 4 * F;
  int slot_i = 0;  // ring slot of the superstage about to be computed
  auto next_slot = [&](int s, int by) { return (s + by) % RS; };
  // the prologue issues what the macro-chunks "before the first" would have: rows(0 .. A) and B(0), B(1) in that order
  if constexpr (A == 1) dma_rows(mc_of(0), 0);
  dma_b(0, 0);
  dma_rows(mc_of(A), A);
  dma_b(1, 1);
  __builtin_amdgcn_s_waitcnt(bits_vmcnt(A ? 2 * NB + RD : NB));  // rows(0) are in their slot
  BH_COMPILER_FENCE();
  uint4 g0[F][4], g1[F][4];
  read_rows(g0, 0);
  auto macro_chunk = [&](int m, uint4 (&g)[F][4], uint4 (&gn)[F][4]) {
    i32x4 bA[NT][4], bB[NT][4];
    // -- even superstage 2 m: pieces 0, 1.  Behind B(2 m) came: [rows(m + A) if A] B(2 m + 1)
    BH_COMPILER_FENCE();
    __builtin_amdgcn_s_waitcnt(bits_vmcnt(NB + A * RD));
    __builtin_amdgcn_s_barrier();                // everybody's share of B(2 m) is in; everybody is done with the slot before
    BH_COMPILER_FENCE();
    dma_b(2 * m + 2, next_slot(slot_i, 2));
    dma_rows(mc_of(m + 1 + A), (m + 1 + A) % SLOTS);  // that row slot is free: macro-chunk m is in registers
    if (const int ss = 2 * mc_of(m); ss >= ss_begin && ss < ss_end) {
      read_b(slot_i, 0, bA);
      read_b(slot_i, 1, bB);
      products(g, std::integral_constant<int, 0>{}, bA);
      products(g, std::integral_constant<int, 1>{}, bB);
    }
    slot_i = next_slot(slot_i, 1);
    // -- odd superstage 2 m + 1: pieces 2, 3.  Behind B(2 m + 1) came: B(2 m + 2), rows(m + 1 + A)
    BH_COMPILER_FENCE();
    __builtin_amdgcn_s_waitcnt(bits_vmcnt(NB + RD));
    __builtin_amdgcn_s_barrier();
    BH_COMPILER_FENCE();
    dma_b(2 * m + 3, next_slot(slot_i, 2));
    const int ss_odd = 2 * mc_of(m) + 1;
    const bool odd_in = ss_odd >= ss_begin && ss_odd < ss_end;
    if (odd_in) {
      read_b(slot_i, 0, bA);
      read_b(slot_i, 1, bB);
      products(g, std::integral_constant<int, 2>{}, bA);
    }
    BH_COMPILER_FENCE();
    // the next macro-chunk's rows are in their slot when at most what was issued behind them is pending:
    //   A = 0: rows(m + 1), B(2 m + 3);   A = 1: rows(m + 1), B(2 m + 1), B(2 m + 2), rows(m + 2), B(2 m + 3)
    __builtin_amdgcn_s_waitcnt(bits_vmcnt(A ? 3 * NB + RD : NB));
    BH_COMPILER_FENCE();
    read_rows(gn, (m + 1) % SLOTS);              // ... and come out of it under the last products of this one
    if (odd_in) products(g, std::integral_constant<int, 3>{}, bB);
    slot_i = next_slot(slot_i, 1);
  };
  int m = 0;
  for (; m + 1 < n_mc; m += 2) {
    macro_chunk(m, g0, g1);
    macro_chunk(m + 1, g1, g0);
  }
  if (m < n_mc) macro_chunk(m, g0, g1);
  BH_COMPILER_FENCE();
  __builtin_amdgcn_s_waitcnt(bits_vmcnt(0));  // the DMAs still on their way land in LDS the way out is about to use
  BH_COMPILER_FENCE();

  gebv_tiles_out<NT, F>(acc, reinterpret_cast<int32_t *>(rowslot), p.out, row0, slice, lane);
}

// ---- the pass as round 3 had it, with round 4's way out: kept for packs of more than five traits --------------------
// Four waves x one 32-row tile (128 rows per workgroup, two workgroups per CU), rows through registers and a turn in
// LDS, the digit operands of a stage through registers into two LDS buffers, a barrier per 128 markers.  The rebuild
// above moves less through LDS per product and needs no registers for staging, and for one column tile (a pass bound by
// its rows) it is the faster one (46.9 against 50.3 us at 8192 x 100k x 1); with two column tiles its longer superstages
// pay more per barrier and the two forms come out 29.3-31.7 against 26-28 us at 2048 x 100k x 8
// (profiles/r04_ab_gebv.txt, section 6) -- so this one serves them.

template <int NT>
__global__ __launch_bounds__(256) void gebv_bits_regs_kernel(const GebvBitsParams p) {
  constexpr int F = 1;
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
  constexpr int kPlaneStrideBits = 144;  // bytes from one row's 128 bytes of planes to the next in LDS
  // (a wave's tile of planes is also where its digit sums are turned on the way out: 32 x 64 int32)
  __shared__ __attribute__((aligned(16))) uint8_t turn[4][32 * 64 * 4];
  static_assert(32 * 64 * 4 >= 32 * kPlaneStrideBits, "the way out's tile holds the planes too");
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
    // (an idle lane takes the row's FIRST piece, whatever the lane: with the lane's own piece offset in the base a
    // row of fewer than eight pieces -- M <= 448 -- would be read past its end)
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

  gebv_tile_out<NT>(acc[0], reinterpret_cast<int32_t *>(&turn[wave][0]), p.out, tile0 * 32, slice, lane);
}

}  // namespace bh
