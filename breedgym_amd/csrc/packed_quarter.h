// The row role of the fused launch for SHORT rows with one trait in the stream (the headline shape: 16 x 256 crosses
// of 10k markers), rebuilt on what packed_dma.h found: a wave's loads queue behind the CU's stores, loads and stores
// share one vmcnt, so the loop must never have to wait for a load that was issued after a store.
//
// Work is cut the other way round.  The old stream gave a wave two whole rows and walked the marker axis, fetching
// three words and 48 bytes of effect digits per lane and vector as it went (123 MB of digit loads per launch out of
// L2, each trip waiting for them behind the previous trip's stores).  Here the four row waves of a workgroup each own a
// QUARTER of the marker axis (NU <= 5 units of 64 words = 512 markers) of the workgroup's eight rows:
//   * the effect digits of its quarter are loaded ONCE, into registers (12 per unit), and serve all eight rows;
//   * the words of its quarter of all eight rows (parent 0, parent 1, mask: 3 x <= 640 B per row) come by LDS-DMA,
//     all 24 instructions issued before the first store;
//   * so every load of the wave is older than every store: the counted wait for row r's words
//     (vmcnt = the DMAs of the later rows + the stores of the earlier ones, exact) never waits for a store, and the
//     wave's <= 48 stores (five 1 KB observation vectors + one packed store per row) all fit vmcnt's 63;
//   * a row's GEBV is the sum of four quarter sums: exact int64 partials meet in LDS (ds_add_u64 + an arrival count,
//     inline asm so that the compiler does not fence them against the DMAs), the last quarter to arrive converts,
//     writes the GEBV and notes the reward key; the wave that finishes the workgroup's last row notes the reward.
// No barrier, the mask waves next door never wait.  NU is a template parameter (the store count of a row is part of the
// wait); a wave whose quarter is ragged (fewer real units) re-stores its last unit to keep the count.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace bh {

constexpr int kQuarterRows = 8;           // rows per workgroup (= four waves x two rows of the old stream: same grid)
constexpr int kQuarterMaxUnits = 5;       // units per wave: rows of up to 4 x 5 x 512 = 10240 markers
constexpr int kQuarterSlotBytes = 3 * kQuarterMaxUnits * 128;           // one row's quarter of the three arrays
constexpr int kQuarterRingBytes = kQuarterRows * kQuarterSlotBytes;     // per row wave: 15 KB

struct QuarterLds {
  unsigned long long acc[kQuarterRows];  // int64 quarter sums of the rows
  unsigned cnt[kQuarterRows];            // quarters arrived
  unsigned best[2];                      // order-preserving keys of the maxima of the first and second env
  unsigned rows_done;
};
__device__ __forceinline__ void quarter_lds_init(QuarterLds *q) {  // thread 0, before the workgroup's barrier
  for (int r = 0; r < kQuarterRows; ++r) {
    q->acc[r] = 0;
    q->cnt[r] = 0;
  }
  q->best[0] = q->best[1] = 0;
  q->rows_done = 0;
}

__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p;
}
__device__ __forceinline__ void lds_add_u64(uint32_t addr, unsigned long long v) {
  asm volatile("ds_add_u64 %0, %1" : : "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t lds_inc_rtn(uint32_t addr) {
  uint32_t old;
  const uint32_t one = 1;
  asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(old) : "v"(addr), "v"(one) : "memory");
  return old;
}
__device__ __forceinline__ unsigned long long lds_read_u64(uint32_t addr) {
  unsigned long long v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t lds_read_u32(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_max_u32(uint32_t addr, uint32_t v) {
  asm volatile("ds_max_u32 %0, %1" : : "v"(addr), "v"(v) : "memory");
}

// the NU words (one per unit) of the three arrays of a row slot: lane l gets word 64 u + l of the quarter
template <int NU>
__device__ __forceinline__ void quarter_read(uint32_t addr, uint32_t (&wa)[NU], uint32_t (&wb)[NU], uint32_t (&wz)[NU]) {
  constexpr int QB = kQuarterMaxUnits * 128;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    asm volatile(
        "ds_read_u16 %0, %3\n\t"
        "ds_read_u16 %1, %3 offset:%4\n\t"
        "ds_read_u16 %2, %3 offset:%5"
        : "=&v"(wa[u]), "=&v"(wb[u]), "=&v"(wz[u])
        : "v"(addr + 128u * (uint32_t)u), "n"(QB), "n"(2 * QB)
        : "memory");
  }
  // one wait behind all of them, tied to every value so that no use is scheduled before it
#pragma unroll
  for (int u = 0; u < NU; ++u) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wa[u]), "+v"(wb[u]), "+v"(wz[u]) : : "memory");
}

// `ring`: kQuarterRingBytes of LDS owned by this wave; `q`: the workgroup's meeting point (zeroed before a barrier).
// MODE_CROSS, one trait (p.t0) in the stream.  p.nvec <= 64 * 4 * NU.
template <int NU, bool OBS>
__device__ __forceinline__ void packed_quarter_role(const PackedParams &p, const int block, const int wave, char *ring,
                                                    QuarterLds *q) {
  static_assert(NU >= 1 && NU <= kQuarterMaxUnits, "units per wave");
  constexpr int ROWS = kQuarterRows, QB = kQuarterMaxUnits * 128;
  constexpr int ST = OBS ? NU + 1 : 1;  // stores per row: NU observation vectors + the packed words of the quarter
  constexpr int DT = 3;
  static_assert(DT * (ROWS - 1) + ST * (ROWS - 1) <= 63, "the counted waits must fit vmcnt's six bits");
  typedef const int32_t __attribute__((address_space(4))) *scalar_i32_ptr;
  typedef short short2v __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63;
  // nothing of this wave is in flight here; said so, the compiler's wait-count model starts clean too (it carried
  // pending loads of the mask role's code, laid out before this one, into waits between the digit loads)
  __builtin_amdgcn_s_waitcnt(vmcnt_only(0));
  const int first = __builtin_amdgcn_readfirstlane(xcd_row(block, p.n_blocks) * ROWS);
  if (first >= p.n_rows) return;
  const int rows_here = p.n_rows - first < ROWS ? p.n_rows - first : ROWS;
  const int units = (p.nvec + 63) >> 6;
  const int u0 = wave * NU;                                                  // this wave's first unit
  const int nu = units - u0 < 0 ? 0 : (units - u0 < NU ? units - u0 : NU);  // real units of this wave (uniform)
  const int env0 = first / p.n_out;
  const uint32_t q_acc = lds_addr(q->acc), q_cnt = lds_addr(q->cnt), q_best = lds_addr(q->best), q_done = lds_addr(&q->rows_done);
  // by a scalar load (not on vmcnt): hipcc answers the use of an ordinary load's result with vmcnt(0) while an LDS-DMA
  // is in flight, which here would be a wait for every store
  const double scale = ((const double __attribute__((address_space(4))) *)(uintptr_t)p.scale)[p.t0];

  // a row's value is known when its fourth quarter has arrived: convert, write, note the reward
  auto arrive = [&](int r, long long sum) {  // lane 0
    lds_add_u64(q_acc + 8u * r, (unsigned long long)sum);
    if (lds_inc_rtn(q_cnt + 4u * r) != 3u) return;
    const int row = first + r;
    const long long total = (long long)lds_read_u64(q_acc + 8u * r);
    const float g = (float)((double)total * scale);
    p.gebv[(int64_t)row * p.T + p.t0] = g;
    if (p.reward) {
      const int env = row / p.n_out;
      if (g == g) {
        if (env - env0 < 2)
          lds_max_u32(q_best + 4u * (env - env0), reward_key(g));
        else
          reward_note(p.reward + (int64_t)env * p.reward_pitch, 0, g);
      }
      if (lds_inc_rtn(q_done) + 1u == (unsigned)rows_here) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const unsigned k = lds_read_u32(q_best + 4u * s);
          if (k) reward_note(p.reward + (int64_t)(env0 + s) * p.reward_pitch, 0, reward_unkey(k));
        }
      }
    }
  };
  if (nu == 0) {  // a quarter beyond the row's end: nothing to move, but the rows wait for four arrivals
    if (lane == 0)
      for (int r = 0; r < rows_here; ++r) arrive(r, 0);
    return;
  }

  const scalar_i32_ptr sidx = (scalar_i32_ptr)(uintptr_t)p.idx;
  const int npieces = p.nvp >> 3;  // 16-byte pieces per packed row
  const int piece0 = 8 * u0;       // this quarter's first piece
  const int my_pieces = npieces - piece0 < 8 * NU ? npieces - piece0 : 8 * NU;  // > 0: nu > 0
  // the words of this quarter of row r -> slot r
  auto dma_row = [&](int r) {
    const int c = first + (r < rows_here ? r : rows_here - 1);  // past the last row that row again: same count
    const int e = c / p.n_out, i = c - e * p.n_out;
    const int i0 = norm_index(sidx[2 * (int64_t)c], p.n_pool), i1 = norm_index(sidx[2 * (int64_t)c + 1], p.n_pool);
    const char *a0 = reinterpret_cast<const char *>(p.src + ((int64_t)e * p.n_pool + i0) * p.nvp);
    const char *a1 = reinterpret_cast<const char *>(p.src + ((int64_t)e * p.n_pool + i1) * p.nvp);
    const char *az = reinterpret_cast<const char *>(p.Z + (int64_t)(p.rng_shared ? i : c) * p.zpitch);
    char *dst = ring + r * kQuarterSlotBytes;
    if (lane < my_pieces) {  // lane 0 always is: the three instructions are issued every time
      const uint32_t off = 16u * (uint32_t)(piece0 + lane);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(a0 + off), (lds_ptr_t)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(a1 + off), (lds_ptr_t)(dst + QB), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(az + off), (lds_ptr_t)(dst + 2 * QB), 16, 0, 0);
    }
  };
  // ---- every load of the wave, oldest first: the effect digits of the quarter, then the rows' words
  const int last = p.nvec - 1;
  // (inline asm for the same reason as the scale: the compiler must not know these loads -- row 0's counted wait is
  // what covers them, they are older than everything it counts)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 dg[NU][3];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    int v = 64 * (u0 + u) + lane;
    v = v < last ? v : last;
    const uint4 *dp = p.dig + (int64_t)p.t0 * 3 * p.nvec + v;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dg[u][j]) : "v"(dp + (int64_t)j * p.nvec) : "memory");
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < ROWS; ++r) dma_row(r);
  __builtin_amdgcn_sched_barrier(0);

  const uint32_t ring_lds = lds_addr(ring);
  // ---- the rows (the row number is part of the wait's immediate: a constant per call)
  auto do_row = [&](auto r_c) {
    constexpr int r = decltype(r_c)::value;
    if (r < rows_here) {
      // behind row r's words: the DMAs of rows r+1 .. 7 and the stores of rows 0 .. r-1 (the digits are older than
      // row 0's words)
      __builtin_amdgcn_s_waitcnt(vmcnt_only(DT * (ROWS - 1 - r) + ST * r));
      if constexpr (r == 0) {  // the digits have landed: from here on they are values like any other
#pragma unroll
        for (int u = 0; u < NU; ++u) asm volatile("" : "+v"(dg[u][0]), "+v"(dg[u][1]), "+v"(dg[u][2]) : : "memory");
      }
      const uint32_t slot = ring_lds + (uint32_t)r * kQuarterSlotBytes;
      uint32_t wa[NU], wb[NU], wz[NU];
      quarter_read<NU>(slot + 2u * (uint32_t)lane, wa, wb, wz);
      uint4 pa, pb, pz;
      {
        const uint32_t a = slot + 16u * (uint32_t)(lane < my_pieces ? lane : 0);
        asm volatile(
            "ds_read_b128 %0, %3\n\t"
            "ds_read_b128 %1, %3 offset:%4\n\t"
            "ds_read_b128 %2, %3 offset:%5\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(pa), "=&v"(pb), "=&v"(pz)
            : "v"(a), "n"(QB), "n"(2 * QB)
            : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      const int row = first + r;
      char *dp = reinterpret_cast<char *>(p.outp + (int64_t)row * p.nvp);
      char *db = OBS ? reinterpret_cast<char *>(p.outb + (int64_t)row * p.pitch) : nullptr;
      // the packed words of the quarter: lane l holds words 8 l .. 8 l + 7 of it (zero past M: the parents are)
      if (lane < my_pieces)
        *reinterpret_cast<uint4 *>(dp + 16u * (uint32_t)(piece0 + lane)) =
            make_uint4(cross32(pa.x, pb.x, pz.x), cross32(pa.y, pb.y, pz.y), cross32(pa.z, pb.z, pz.z), cross32(pa.w, pb.w, pz.w));
      int acc[3] = {0, 0, 0};
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        // a unit past the quarter's end (ragged quarter) redoes the last real one: the same store again, not counted
        const int uu = u < nu ? u : nu - 1;
        const int w = 64 * (u0 + uu) + lane;
        const bool has = u < nu && w < p.nvec;
        const int wc = w < last ? w : last;  // a lane past the row's end redoes the last vector
        uint32_t o = cross16(wa[u], wb[u], wz[u]);
        if (u >= nu || w > last) {  // take the words of the vector that is redone
          const uint32_t src = slot + 2u * (uint32_t)(wc - 64 * u0);
          uint32_t xa, xb, xz;
          asm volatile(
              "ds_read_u16 %0, %3\n\t"
              "ds_read_u16 %1, %3 offset:%4\n\t"
              "ds_read_u16 %2, %3 offset:%5\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(xa), "=&v"(xb), "=&v"(xz)
              : "v"(src), "n"(QB), "n"(2 * QB)
              : "memory");
          o = cross16(xa, xb, xz);
        }
        const uint4 oo = expand_vector(o);
        if constexpr (OBS) *reinterpret_cast<uint4 *>(db + 16u * (uint32_t)wc) = oo;
        if (has) {
          const uint32_t ww[4] = {oo.x, oo.y, oo.z, oo.w};
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const uint32_t dd[4] = {dg[u][j].x, dg[u][j].y, dg[u][j].z, dg[u][j].w};
            int a = acc[j];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const uint32_t pr = (ww[k] + (ww[k] >> 8)) & 0x00FF00FFu;  // dosages of a marker pair as two int16
              a = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, pr), __builtin_bit_cast(short2v, dd[k]), a, false);
            }
            acc[j] = a;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // digits recombined in int64 per lane, summed over the wave, met with the other quarters in LDS: exact, order-free
      long long sum = (long long)acc[0] + (long long)acc[1] * 65536LL + (long long)acc[2] * 4294967296LL;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
      if (lane == 0) arrive(r, sum);
    }
  };
  static_assert(ROWS == 8, "unrolled by hand");
  do_row(std::integral_constant<int, 0>{});
  do_row(std::integral_constant<int, 1>{});
  do_row(std::integral_constant<int, 2>{});
  do_row(std::integral_constant<int, 3>{});
  do_row(std::integral_constant<int, 4>{});
  do_row(std::integral_constant<int, 5>{});
  do_row(std::integral_constant<int, 6>{});
  do_row(std::integral_constant<int, 7>{});
}

}  // namespace bh
