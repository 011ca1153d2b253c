// Bit-packed genotype rows: 1 bit per allele, 8x less to read than the byte layout of the API.
//
// A packed row is uint16 P[8 * nw] (nw = ceil(M / 64)): bit i of P[v] is byte i of the 16-byte vector
// v of the byte-layout row -- markers 8v .. 8v+7, even bits their allele 0, odd bits their allele 1
// (numpy: packbits(row_bytes, bitorder="little") viewed as uint16); bits of markers >= M are zero.
// That is also the shape of the crossover-mask rows the mask role writes (even bits: gamete 0 = the
// mask for parent 0, odd bits: gamete 1), and P[v] is exactly what one lane needs to produce vector v
// -- so the row role of a cross becomes
//     three 2-byte loads (parent 0, parent 1, mask)  ->  8+8 bit selects  ->  2-byte store
//     [-> 16-byte coalesced store of the byte-layout offspring, if the caller wants observations]
//     [-> GEBV from the expanded vector]
// Per cross that is M/2 + M/4 bytes read and M/4 (+ 2M) written instead of 4M read and 2M written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "founders.h"
#include "mask_stream.h"
#include "row_kernels.h"

namespace bh {

constexpr int kLongRowPrio = 2;  // priority of the row waves of the long-row stream
constexpr int kLongRowVectors = 4;   // vectors per lane and trip in the long-row stream
constexpr int kShortRowVectors = 2;  // ... and in the stream with GEBV
// ... per number of traits in the stream: two traits hold twice the digit registers -- with two vectors per trip the
// two-role kernel needs 130 VGPRs (one workgroup per CU), with one 76: 26.9 against 29.4 us per step at 16 x 256 x 10k
template <int TT> constexpr int obs_vectors() { return TT == 0 ? kLongRowVectors : TT == 2 ? 1 : kShortRowVectors; }
constexpr int kNoObsVectors = 1;     // ... with GEBV but without the byte observation (2: the same; 4: +6 us per launch)
constexpr int kLongAhead = 1, kShortAhead = 1;  // trips the stream requests ahead: without / with GEBV

// the sum of `v` over the wave's 64 lanes (it forms in lane 63 and is handed to all): four DPP steps
// inside a row of 16 lanes (two quad swaps, the half mirror, the mirror), two across rows (row_bcast15, row_bcast31)
// -- no LDS round trips (a 64-bit __shfl_xor reduction is twelve ds_bpermute with a wait after every pair)
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2, 3, 0, 1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);  // row_mirror: every lane of a row has the row's sum
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}

struct PackedParams {
  const uint16_t *src;  // cross: (E, n_pool, nvp) packed pool; gather: (n_pool, nvp)
  uint16_t *outp;       // (rows, nvp) packed rows out, or null
  uint8_t *outb;        // (rows, pitch) byte-layout rows out, or null
  const int32_t *idx;   // cross: (rows, 2); gather: (rows) or null (= identity)
  float *gebv;          // (rows, T) or null
  const uint8_t *Z;     // cross: (B, zpitch) masks from the mask role
  const double *eff;    // (tcount, effPitch) f64 marker effects of traits t0.., zero padded
  const uint4 *dig;     // (T, 3, nvec) 16-byte blocks: digit j of trait t for the 8 markers of vector v, as int16
  const double *scale;  // (T) 2^-s_t of the fixed-point effects
  int64_t pitch;        // bytes per byte-layout row
  int64_t effPitch;
  int64_t zpitch;       // bytes per mask row = bytes per packed row = 16 * pw (ctx.h)
  int32_t nvec;         // 16-byte vectors per byte-layout row = ceil(M / 8)
  int32_t nvp;          // uint16 words per packed row = 8 * pw: nvec <= nvp < nvec + 64, words nvec .. nvp - 1 zero
  int32_t n_pool, n_out;
  int32_t n_rows;
  int32_t n_blocks;
  int32_t T, t0, tcount;
  int32_t rng_shared;
  int32_t seq;  // row groups a wave of packed_stream_role streams one after the other (>= 1)
  float *reward;       // cross: (n_envs) max GEBV per env, holding -inf on entry (reward_note), or null
  float *reward_next;  // (n_envs) set to -inf by this launch for a later one, or null
  int64_t reward_pitch;  // floats between the words of two envs in both (RowParams)
  int32_t n_envs;
  int32_t dig_lds;  // 1 + byte offset, in the launch's dynamic LDS, of a copy of trait 0's three digit planes that the
                    // row waves of a workgroup make and read (packed_stream_role, DIGLDS); 0 = digits from memory
  int32_t obs_nt;  // how the pipelined stream stores the byte observation: 0 write-through (sc1), 1 non-temporal
                   // (obs_store_policy in cross.hip)
  uint32_t *err;  // the context's device-side error word (host memory): a wait among a workgroup's waves that ran out
  char *spare;  // 16 KB nobody reads: where the lanes past a row's end send their stores in the LDS-DMA stream
                // (packed_dma.h), so that every trip issues the same number of memory operations
};

// dword k of byte-layout vector v from its packed word: bits 4k .. 4k+3 are its four bytes; the
// multiply drops them onto bit 0 of bytes 0, 1, 2, 3.
__device__ __forceinline__ uint32_t expand_dword(uint32_t o16, int k) {
  return (((o16 >> (4 * k)) & 0xfu) * 0x00204081u) & 0x01010101u;
}

__device__ __forceinline__ uint4 expand_vector(uint32_t o16) {
  return make_uint4(expand_dword(o16, 0), expand_dword(o16, 1), expand_dword(o16, 2), expand_dword(o16, 3));
}

// The offspring word of a cross, 8 markers at once.  Allele 0 of the offspring (even bit) is the gamete of
// parent 0: its allele-0 bit where the mask bit is 0, its allele-1 bit where it is 1; allele 1 (odd bit) is
// the gamete of parent 1 under the odd mask bits.  Three bit-field inserts: at the even positions of
// bfi(z, p0 >> 1, p0) the even mask bits choose between p0's odd and even bits, at the odd positions of
// bfi(z, p1, p1 << 1) the odd mask bits choose between p1's odd and even bits; the third keeps each half.
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (mask & a) | (~mask & b); }
__device__ __forceinline__ uint32_t cross16(uint32_t p0, uint32_t p1, uint32_t z) {
  return bfi(0x5555u, bfi(z, p0 >> 1, p0), bfi(z, p1, p1 << 1)) & 0xffffu;
}

// One wave streams R consecutive rows (the marker effects of a trip are loaded once for all of
// them, as in row_role); MODE_CROSS or MODE_GATHER.  The loop is deliberately plain: requesting the
// next trips' words ahead of the current trip's stores (software pipelining) was measured -- it costs
// registers the mask role next door needs more than the row role needs the overlap (docs/NOTEBOOK.md 4b).
template <int TT, int MODE, int R, int PRIO = 3>
__device__ __forceinline__ void packed_row_role(const PackedParams &p, const int block, const int wave,
                                                RewardLds *rw = nullptr) {
  static_assert(MODE == MODE_CROSS || MODE == MODE_GATHER, "packed rows: cross or gather");
  constexpr int NT = TT > 0 ? TT : 1;
  const int lane = threadIdx.x & 63;
  const int c0 = __builtin_amdgcn_readfirstlane((xcd_row(block, p.n_blocks) * 4 + wave) * R);
  if (c0 >= p.n_rows) return;

  const uint16_t *src0[R], *src1[R], *Z16[R];
  uint16_t *dstp[R];
  uint8_t *dstb[R];
  bool live[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    live[r] = c0 + r < p.n_rows;
    const int c = live[r] ? c0 + r : c0;  // a dead slot re-reads row c0 and writes nothing
    src1[r] = nullptr;
    Z16[r] = nullptr;
    if constexpr (MODE == MODE_CROSS) {
      const int e = c / p.n_out, i = c - e * p.n_out;
      const int i0 = norm_index(p.idx[2 * (int64_t)c], p.n_pool);
      const int i1 = norm_index(p.idx[2 * (int64_t)c + 1], p.n_pool);
      src0[r] = p.src + ((int64_t)e * p.n_pool + i0) * p.nvp;
      src1[r] = p.src + ((int64_t)e * p.n_pool + i1) * p.nvp;
      Z16[r] = reinterpret_cast<const uint16_t *>(p.Z + (int64_t)(p.rng_shared ? i : c) * p.zpitch);
    } else {
      src0[r] = p.src + (int64_t)(p.idx ? norm_index(p.idx[c], p.n_pool) : c) * p.nvp;
    }
    dstp[r] = (p.outp && live[r]) ? p.outp + (int64_t)c * p.nvp : nullptr;
    dstb[r] = (p.outb && live[r]) ? p.outb + (int64_t)c * p.pitch : nullptr;
  }
  if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);  // see row_role

  double acc[R][NT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[r][t] = 0.0;

  for (int v = lane; v < p.nvp; v += 64) {
    uint32_t a[R], b[R], z[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      a[r] = src0[r][v];
      if constexpr (MODE == MODE_CROSS) {
        b[r] = src1[r][v];
        z[r] = Z16[r][v];
      }
    }
    const bool inrow = v < p.nvec;  // the packed row is padded to whole 64-marker words
    uint4 oo[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      uint32_t o = a[r];
      if constexpr (MODE == MODE_CROSS) o = cross16(a[r], b[r], z[r]);
      if (dstp[r]) dstp[r][v] = (uint16_t)o;
      oo[r] = expand_vector(o);
      if (dstb[r] && inrow) *reinterpret_cast<uint4 *>(dstb[r] + 16 * (int64_t)v) = oo[r];
    }
    if constexpr (TT > 0) {
      if (inrow) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          if (t < p.tcount) {
            const double2 *ep = reinterpret_cast<const double2 *>(p.eff + (int64_t)t * p.effPitch + 8 * (int64_t)v);
            const double2 e0 = ep[0], e1 = ep[1], e2 = ep[2], e3 = ep[3];
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const Dosage8 d = dosages(oo[r]);
              double s = acc[r][t];
              s = fma(d.d[0], e0.x, s);
              s = fma(d.d[1], e0.y, s);
              s = fma(d.d[2], e1.x, s);
              s = fma(d.d[3], e1.y, s);
              s = fma(d.d[4], e2.x, s);
              s = fma(d.d[5], e2.y, s);
              s = fma(d.d[6], e3.x, s);
              s = fma(d.d[7], e3.y, s);
              acc[r][t] = s;
            }
          }
        }
      }
    }
  }

  if constexpr (TT > 0) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        double s = acc[r][t];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        acc[r][t] = s;
      }
    if (lane == 0) {
      int env0 = 0, expected = 0;
      if constexpr (MODE == MODE_CROSS)
        if (p.reward) reward_wg_layout(block, p.n_blocks, R, p.n_rows, p.n_out, env0, expected);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (live[r]) {
          float best = -INFINITY;
#pragma unroll
          for (int t = 0; t < TT; ++t)
            if (t < p.tcount) {
              const float g = (float)acc[r][t];
              p.gebv[(int64_t)(c0 + r) * p.T + p.t0 + t] = g;
              best = g != g ? best : fmaxf(best, g);
            }
          if constexpr (MODE == MODE_CROSS)
            if (p.reward) reward_row(rw, p.reward, p.reward_pitch, env0, c0 + r, p.n_out, best);
        }
      if constexpr (MODE == MODE_CROSS)
        if (p.reward) reward_wave_done(rw, p.reward, p.reward_pitch, env0, expected);
    }
  }
}

// The row role of the fused launch next to four-wave mask groups, and the stream of the standalone
// launches that write packed rows (reset's founder gather), software-pipelined like row_role_pipelined
// (row_kernels.h): the 2-byte words of trip t+1 are requested before trip t's stores and the marker
// effects of trip t before those, no wait covers a store, the body is branch-free (lanes past the end of
// the row redo the last vector, dead row slots redo row c0; the few padding words of a packed row are
// written behind the loop), no priority raise.
// TT = 0: no GEBV in the stream (long rows: the matrix-core pass over the packed offspring follows,
// gebv_bits.h).  p.seq row groups per wave, one after the other: that keeps the grid of a launch of many
// long rows small enough to be resident at once (not a question of DRAM pages: stores run at the same rate for
// 512 ... 8192 concurrent row streams, tools/store_streams.hip, docs/NOTEBOOK.md section 4c.3).
// WPB: waves of a workgroup that stream rows (four beside four mask waves; all eight in the workgroups of a gather that
// rides on a cross launch, packed_cross_kernel).
// DIGLDS (one trait): the digit planes come from a copy in LDS that the workgroup's WPB row waves make when they enter
// -- each a share, by LDS-DMA, under the two round trips to memory every wave starts with (indices, first words) -- and
// meet over an LDS counter before the first trip.  From memory every wave re-read the planes for its R rows: 123 MB
// through the CUs' L1s per 16 x 256 x 10k launch, more than the observation it stores, on the path its stores take
// (launch 21.3-21.5 -> 19.7-19.8 us on one box; with no digit loads at all it would be 18.1:
// profiles/r04_ab_digits_lds.txt).
template <int TT, int MODE, int R, bool OBS, int VPT = 1, bool OBSNT = false, int WPB = 4, bool DIGLDS = false>
__device__ __forceinline__ void packed_stream_role(const PackedParams &p, const int block, const int wave,
                                                   RewardLds *rw = nullptr, char *dig_lds = nullptr,
                                                   unsigned *dig_ready = nullptr) {
  static_assert(!DIGLDS || TT == 1, "the LDS copy holds one trait's digit planes");
  static_assert(TT >= 0 && TT <= 2, "effects are held in registers across the trip");
  static_assert(MODE == MODE_CROSS || MODE == MODE_GATHER, "packed rows: cross or gather");
  constexpr int NT = TT > 0 ? TT : 1;
  const int lane = threadIdx.x & 63;
  const int seq = p.seq > 1 ? p.seq : 1;
  // without GEBV the stream is almost all memory instructions: above the mask waves it keeps the memory pipeline
  // fed and they take every issue slot it leaves (their role has no vector loads to be starved of, mask_stream.h)
  if constexpr (TT == 0) __builtin_amdgcn_s_setprio(kLongRowPrio);
  static_assert(WPB == 4 || MODE == MODE_GATHER, "the reward's workgroup layout counts four row waves");
  const int first = __builtin_amdgcn_readfirstlane((xcd_row(block, p.n_blocks) * WPB + wave) * R * seq);
  // the traits' scales are asked for here: read where they are used -- after the last trip, by one lane -- they cost
  // every wave a memory round trip at the very end of its life (a microsecond of the configs[1] launch)
  double scale[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) scale[t] = TT > 0 ? p.scale[p.t0 + t] : 0.0;
  // this wave's share of the digit copy: KB chunks wave, wave + WPB, ... (64 pieces of 16 bytes each; the last chunk's
  // spare lanes fetch the last piece again -- the copy's LDS is whole KBs).  Asked for BEHIND the first words: a CU's
  // memory pipeline is in order, and in front of them the 15 KB per wave kept every row's first store waiting
  auto fill_digits = [&]() {
    typedef __attribute__((address_space(3))) void *lds_p;
    typedef const __attribute__((address_space(1))) void *glb_p;
    const int npieces = 3 * (p.nvec + 1);
    const char *g = reinterpret_cast<const char *>(p.dig);
    for (int ch = wave; 64 * ch < npieces; ch += WPB) {
      const int piece = 64 * ch + lane < npieces ? 64 * ch + lane : npieces - 1;
      __builtin_amdgcn_global_load_lds((glb_p)(g + 16u * (uint32_t)piece), (lds_p)(dig_lds + 1024 * ch), 16, 0, 0);
    }
  };
  if constexpr (DIGLDS) {
    if (first >= p.n_rows) {  // a wave without rows still owes the others its share
      fill_digits();
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      if (lane == 0) __hip_atomic_fetch_add(dig_ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      return;
    }
  }
  for (int sq = 0; sq < seq; ++sq) {
    const int c0 = first + sq * R;
    if (c0 >= p.n_rows) return;

    const uint16_t *src0[R], *src1[R], *Z16[R];
    uint16_t *dstp[R];
    uint8_t *dstb[R];
    bool live[R];
    typedef const int32_t __attribute__((address_space(4))) *scalar_i32_ptr;
    const scalar_i32_ptr sidx = (scalar_i32_ptr)(uintptr_t)p.idx;
    int raw0[R], raw1[R];  // the rows' parent indices as the caller wrote them, all asked for before any is used
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int c = c0 + r < p.n_rows ? c0 + r : c0;
      raw0[r] = raw1[r] = c;
      if constexpr (MODE == MODE_CROSS) {
        raw0[r] = sidx[2 * (int64_t)c];
        raw1[r] = sidx[2 * (int64_t)c + 1];
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      live[r] = c0 + r < p.n_rows;
      const int c = live[r] ? c0 + r : c0;
      src1[r] = nullptr;
      Z16[r] = nullptr;
      // the parents' indices are wave-uniform; said so, every row base is a scalar and the lanes address
      // with one 32-bit offset (no 64-bit address arithmetic, a third of the registers)
      if constexpr (MODE == MODE_CROSS) {
        const int e = c / p.n_out, i = c - e * p.n_out;
        // ... and read through the scalar cache (as packed_dma.h reads them): the pairs of the wave's R rows are
        // asked for together and come as scalars; as vector loads each row's pair was waited for before the next
        // row's was asked for -- a memory round trip per row before the first word load
        const int i0 = norm_index(raw0[r], p.n_pool), i1 = norm_index(raw1[r], p.n_pool);
        src0[r] = p.src + ((int64_t)e * p.n_pool + i0) * p.nvp;
        src1[r] = p.src + ((int64_t)e * p.n_pool + i1) * p.nvp;
        Z16[r] = reinterpret_cast<const uint16_t *>(p.Z + (int64_t)(p.rng_shared ? i : c) * p.zpitch);
      } else {
        const int i0 = p.idx ? __builtin_amdgcn_readfirstlane(norm_index(p.idx[c], p.n_pool)) : c;
        src0[r] = p.src + (int64_t)i0 * p.nvp;
      }
      dstp[r] = p.outp + (int64_t)c * p.nvp;
      dstb[r] = OBS ? p.outb + (int64_t)c * p.pitch : nullptr;
    }
    __amdgpu_buffer_rsrc_t obs_rsrc[R];  // the rows' observation blocks (wave-uniform bases)
#pragma unroll
    for (int r = 0; r < R; ++r)
      obs_rsrc[r] = __builtin_amdgcn_make_buffer_rsrc((void *)dstb[r], 0, OBS ? (int)p.pitch : 0, 0x00020000);
    auto word_at = [](const uint16_t *base, uint32_t v) -> uint32_t {
      return *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(base) + 2u * v);
    };

    // GEBV in exact fixed point: the effects are round(e * 2^s_t) cut into three balanced base-65536
    // digits (as gebv_mfma.h cuts them into six base-256 ones), the dosages of a marker pair are the
    // two int16 halves of (w + (w >> 8)) & 0x00FF00FF, and a v_dot2_i32_i16 per pair and digit replaces
    // eight conversions and eight f64 FMAs per vector.  |digit| <= 2^15 and dosage <= 2, so a lane's
    // int32 sums stay within int32 for rows of up to 2^21 markers = bh_max_markers().
    int acc[R][NT][3];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[r][t][j] = 0;

    const int last = p.nvec - 1;
    // VPT vectors per lane and trip (64 apart: every load and store of the wave stays contiguous): with few
    // waves per row stream the bytes in flight have to come from each wave
    struct Words {
      uint32_t a[VPT][R], b[VPT][R], z[VPT][R];
    };
    auto load_words = [&](int base, Words &w) {
#pragma unroll
      for (int u = 0; u < VPT; ++u) {
        int v = base + 64 * u + lane;
        v = v < last ? v : last;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          w.a[u][r] = word_at(src0[r], (uint32_t)v);
          if constexpr (MODE == MODE_CROSS) {
            w.b[u][r] = word_at(src1[r], (uint32_t)v);
            w.z[u][r] = word_at(Z16[r], (uint32_t)v);
          } else {
            w.b[u][r] = 0;
            w.z[u][r] = 0;
          }
        }
      }
    };
    constexpr int AHEAD = TT == 0 ? kLongAhead : kShortAhead;
    const char *plane[NT][3];  // the digit planes' bases: scalars, so that every digit load is base + one 32-bit offset
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        plane[q][j] = reinterpret_cast<const char *>(p.dig + ((int64_t)(p.t0 + q) * 3 + j) * ((int64_t)p.nvec + 1));
      }
    const __attribute__((address_space(3))) char *dig3 = (const __attribute__((address_space(3))) char *)dig_lds;
    auto trip = [&](int base, const Words &w, Words &next) {
      const int v = base + lane;
      uint4 dg[VPT][NT][3];
      if constexpr (TT > 0) {
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
          // a lane that redoes the row's last vector must not count it twice: it reads the zero block that ends every
          // digit plane.  Plane bases are wave-uniform, the lanes address with one 32-bit offset.
          const uint32_t vd = (uint32_t)(v + 64 * u < p.nvec ? v + 64 * u : p.nvec);
#pragma unroll
          for (int q = 0; q < TT; ++q)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              if constexpr (DIGLDS)  // (an LDS pointer by type, not by inference: never a flat or global load)
              {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 d = *reinterpret_cast<const __attribute__((address_space(3))) u32x4 *>(
                    dig3 + 16u * (uint32_t)j * (uint32_t)(p.nvec + 1) + 16u * vd);
                dg[u][q][j] = make_uint4(d[0], d[1], d[2], d[3]);
              }
              else
                dg[u][q][j] = *reinterpret_cast<const uint4 *>(plane[q][j] + 16u * vd);
            }
        }
      }
      load_words(base + 64 * VPT * AHEAD, next);
      typedef short short2v __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
          const int vu = v + 64 * u < last ? v + 64 * u : last;
          uint32_t o = w.a[u][r];
          if constexpr (MODE == MODE_CROSS) o = cross16(w.a[u][r], w.b[u][r], w.z[u][r]);
          *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(dstp[r]) + 2u * (uint32_t)vu) = (uint16_t)o;
          const uint4 oo = expand_vector(o);
          if constexpr (OBS) {
            // 16-byte stores through the row's buffer descriptor, so that the cache policy is an immediate the
            // compiler knows (its waits stay counted): write-through (sc1, aux 16) or non-temporal (nt, aux 2)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 dv = {oo.x, oo.y, oo.z, oo.w};
            __builtin_amdgcn_raw_buffer_store_b128(dv, obs_rsrc[r], (int)(16u * (uint32_t)vu), 0, OBSNT ? 2 : 16);
          }
          if constexpr (TT > 0) {
            // (dosage of marker 2k, dosage of marker 2k+1) as two int16; a lane that redoes the row's last vector
            // multiplies them by the zero block of the digit planes
            const uint32_t ww[4] = {oo.x, oo.y, oo.z, oo.w};
            uint32_t pr[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) pr[k] = (ww[k] + (ww[k] >> 8)) & 0x00FF00FFu;
#pragma unroll
            for (int q = 0; q < TT; ++q)
#pragma unroll
              for (int j = 0; j < 3; ++j) {
                const uint32_t dd[4] = {dg[u][q][j].x, dg[u][q][j].y, dg[u][q][j].z, dg[u][q][j].w};
                int a = acc[r][q][j];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                  a = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, pr[k]), __builtin_bit_cast(short2v, dd[k]), a, false);
                acc[r][q][j] = a;
              }
          }
        }
    };
    if constexpr (AHEAD == 1) {
      Words wa, wb;
      load_words(0, wa);
      if constexpr (DIGLDS)
        if (sq == 0) fill_digits();
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): enter the loop with nothing pending (row_role_pipelined)
      if constexpr (DIGLDS) {
        if (sq == 0) {  // this wave's share of the digit copy has landed (the wait above); the others'?
          if (lane == 0) __hip_atomic_fetch_add(dig_ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          // (bounded: the waves it waits for are resident and owe nothing to this one; a wait that runs out is
          // reported through the context's error word, like the mask role's hand-off: check_ctx)
          bool ok = false;
          for (int spin = 0; spin < (1 << 22); ++spin) {
            if (__hip_atomic_load(dig_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= (unsigned)WPB) {
              ok = true;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
          if (!ok && lane == 0 && p.err) __hip_atomic_fetch_or(p.err, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      for (int base = 0; base < p.nvec; base += 128 * VPT) {  // two trips per turn, no branch between them
        trip(base, wa, wb);
        trip(base + 64 * VPT, wb, wa);
      }
    } else {  // two trips ahead: three register sets in rotation (trip t fills the set trip t + 2 reads)
      Words wa, wb, wc;
      load_words(0, wa);
      load_words(64 * VPT, wb);
      for (int base = 0; base < p.nvec; base += 192 * VPT) {
        trip(base, wa, wc);
        trip(base + 64 * VPT, wb, wa);
        trip(base + 128 * VPT, wc, wb);
      }
    }
    if (p.nvec + lane < p.nvp) {  // padding words: markers beyond M, up to the pitch (fewer than 64: ctx.h, pw)
#pragma unroll
      for (int r = 0; r < R; ++r) dstp[r][p.nvec + lane] = 0;
    }

    if constexpr (TT > 0) {
      // (with GEBV in the stream a wave has one row group: seq == 1)
      int env0 = 0, expected = 0;
      if constexpr (MODE == MODE_CROSS)
        if (p.reward) reward_wg_layout(block, p.n_blocks, R, p.n_rows, p.n_out, env0, expected);
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          // exact, order-free: the three digit sums are added over the wave, then recombined in int64.  Rows of up to
          // 16384 markers (every row that has its GEBV made here when there are 32 rows or more: cross.hip,
          // use_bits_gebv) keep a wave's digit sum inside an int32 -- 2 * 32768 * 16384 = 2^30 -- and add it with six
          // DPP steps per digit; longer rows recombine per lane first and go through the 64-bit shuffle.
          long long sum;
          if (p.nvec <= 2048) {
            sum = (long long)wave_sum_i32(acc[r][t][0]) + (long long)wave_sum_i32(acc[r][t][1]) * 65536LL +
                  (long long)wave_sum_i32(acc[r][t][2]) * 4294967296LL;
          } else {
            sum = (long long)acc[r][t][0] + (long long)acc[r][t][1] * 65536LL + (long long)acc[r][t][2] * 4294967296LL;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
          }
          if (lane == 0 && live[r]) {
            const float g = (float)((double)sum * scale[t]);
            p.gebv[(int64_t)(c0 + r) * p.T + p.t0 + t] = g;
            if constexpr (MODE == MODE_CROSS)
              if (p.reward) reward_row(rw, p.reward, p.reward_pitch, env0, c0 + r, p.n_out, g);
          }
        }
      if constexpr (MODE == MODE_CROSS)
        if (lane == 0 && p.reward) reward_wave_done(rw, p.reward, p.reward_pitch, env0, expected);
    }
  }
}

}  // namespace bh
#include "packed_dma.h"
namespace bh {

template <int TT, int MODE, int R>
__global__ __launch_bounds__(256) void packed_row_kernel(const PackedParams p) {
  __shared__ RewardLds rwl;
  prime_next_reward(p.reward_next, p.n_envs, p.reward_pitch);
  if (p.reward) {
    if (threadIdx.x == 0) reward_lds_init(&rwl);
    __syncthreads();
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: keep it scalar
  if constexpr (TT == 0 && R == 1) {
    // long rows in turns (the founder gather of a reset, a cross without run-ahead): the LDS-DMA stream
    if (p.outp && p.seq > 1) {
      __shared__ __attribute__((aligned(16))) char ring[4][MODE == MODE_CROSS ? kDmaRingBytes : kDmaGatherRingBytes];
      if (p.outb)
        packed_dma_role<MODE, true>(p, (int)blockIdx.x, wave, ring[wave]);
      else
        packed_dma_role<MODE, false>(p, (int)blockIdx.x, wave, ring[wave]);
      return;
    }
  }
  if constexpr (TT <= 2) {
    // packed rows out (reset's founder gather, a cross without run-ahead): the pipelined stream; without
    // GEBV only where the host asks for it (p.seq >= 1: long rows, in turns when there are many)
    if (p.outp && (TT > 0 || p.seq >= 1)) {
      // with GEBV in the stream: two vectors per lane and trip when the observation is written too (-2.5 % at
      // 16 x 256 x 10k), one otherwise
      if (p.outb && p.obs_nt)
        packed_stream_role<TT, MODE, R, true, obs_vectors<TT>(), true>(p, (int)blockIdx.x, wave, &rwl);
      else if (p.outb)
        packed_stream_role<TT, MODE, R, true, obs_vectors<TT>(), false>(p, (int)blockIdx.x, wave, &rwl);
      else
        packed_stream_role<TT, MODE, R, false, TT == 0 ? kLongRowVectors : kNoObsVectors>(p, (int)blockIdx.x, wave, &rwl);
      return;
    }
  }
  packed_row_role<TT, MODE, R>(p, (int)blockIdx.x, wave, &rwl);
}

// cross_kernel (row_kernels.h) with the packed row role: waves 0-3 stream generation g, waves 4-7
// make the masks of generation g+1.
// fp.out != null (MW = 4 only): the last fp.E workgroups of the grid draw the founders of the reset that follows
// this step (founders_role): 12 us of LDS sorting by 16 workgroups that would otherwise be a launch of its own
// between the horizon step and the founder gather.
// RIDE: the instance that can carry a gather asked for ahead (gp); the steps of an episode run the one that cannot --
// fewer registers (with the riders' paths inlined the two-trait instance went from 124 to 131 VGPRs: one workgroup
// per CU instead of two).
// LAYOUT: the Threefry counter layout of the mask role, an instance per layout (mask_role) -- or -1 for the instances
// whose mask role draws one gamete stream per lane (mp.sites: the long-row launch of few keyed crosses).
template <int TT, int MW, int R, bool RIDE = false, int LAYOUT = -1>
__global__ __launch_bounds__(512) void packed_cross_kernel(const PackedParams rp, const MaskParams mp,
                                                          const FoundersParams fp, const PackedParams gp) {
  static_assert(!RIDE || (MW == 4 && TT >= 1 && TT <= 2), "a gather rides on four-wave mask groups with fused GEBV");
  extern __shared__ __align__(16) uint8_t lds_raw[];
  __shared__ unsigned counter, done;
  __shared__ RewardLds rwl;
  __shared__ unsigned dig_ready;  // row waves whose share of the digit copy has landed (packed_stream_role, DIGLDS)
  constexpr bool RIDERS = RIDE;
  int n_main = (int)gridDim.x;  // two-role workgroups
  int block = (int)blockIdx.x;  // ... and which of them this is
  bool riding = false;          // a workgroup of the gather that rides on this launch (gp)
  if constexpr (MW == 4) {
    // fp.out != null: fp.E workgroups of the grid draw the founders of a reset to come (founders_role: 12 us of LDS
    // sorting that would otherwise be a launch of its own).  They are the last of the grid -- the row work is not held
    // up -- unless a gather rides too: that launch runs its workgroups in two rounds, and a draw that starts with the
    // first ends long before the second does
    const int nf = fp.out ? fp.E : 0;
    const bool f_first = RIDERS && gp.n_blocks > 0;
    n_main -= nf;
    if (nf > 0 && (f_first ? block < nf : block >= n_main)) {  // uniform per workgroup
      founders_role(fp, f_first ? block : block - n_main, reinterpret_cast<uint64_t *>(lds_raw));
      return;
    }
    if (f_first) block -= nf;
    if constexpr (RIDERS) {
      // gp.n_blocks > 0: a gather asked for ahead (bh_gather_packed_deferred: the founders of the reset that follows
      // this horizon step) rides on this launch as gp.n_blocks more workgroups, all eight waves of which stream
      // gathered rows (bit planes, byte observation, GEBV).  Eights of two-role workgroups and eights of riders
      // alternate for as long as both last: the riders' stores (the observation) flow from the first microsecond
      // beside the crosses, which at a horizon write no observation, and blockIdx % 8 -- the XCD -- stays what xcd_row
      // expects of a workgroup's number.  At 16 x 256 x 10k the launch takes 31 us (a horizon launch 17.8 + a gather
      // launch 18 + the gap between them before): its 784 workgroups are two rounds of latency chains (a wave's rows
      // are ten trips, each a round trip to memory), whichever way the riders are cut -- four row waves in two rounds
      // 31.7, half as many with two row groups each 31.7, eight row waves 31.2 (profiles/r04_fused_reset.txt).
      if (gp.n_blocks > 0) {
        n_main -= gp.n_blocks;
        const int both = (n_main < gp.n_blocks ? n_main : gp.n_blocks) & ~7;
        if (block < 2 * both) {
          riding = (block >> 3) & 1;
          block = ((block >> 4) << 3) | (block & 7);
        } else {
          const int rest = block - 2 * both;
          riding = rest >= n_main - both;
          block = both + (riding ? rest - (n_main - both) : rest);
        }
      }
    }
  }
  prime_next_reward(rp.reward_next, rp.n_envs, rp.reward_pitch, block == 0 && !riding);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: keep it scalar
  if (MW == 4 || rp.reward) {  // uniform: the whole workgroup meets here or nowhere
    if (threadIdx.x == 0) {
      counter = 0;
      done = 0;
      dig_ready = 0;
      reward_lds_init(&rwl);
    }
    __syncthreads();
  }
  if constexpr (RIDERS) {
    if (riding) {  // all eight waves stream rows: a rider has no mask work
      if constexpr (TT == 1) {
        if (gp.dig_lds && gp.outb && !gp.obs_nt) {
          packed_stream_role<TT, MODE_GATHER, R, true, kShortRowVectors, false, 8, true>(
              gp, block, wave, &rwl, reinterpret_cast<char *>(lds_raw) + (gp.dig_lds - 1), &dig_ready);
          return;
        }
      }
      if (gp.outb)  // (write-through: a gather whose observation wants non-temporal stores does not ride, cross.hip)
        packed_stream_role<TT, MODE_GATHER, R, true, obs_vectors<TT>(), false, 8>(gp, block, wave, &rwl);
      else
        packed_stream_role<TT, MODE_GATHER, R, false, 1, false, 8>(gp, block, wave, &rwl);
      return;
    }
  }
  if (wave < 4) {
    if (block < rp.n_blocks) {
      if constexpr (TT <= 2 && MW == 4) {
        if (TT > 0 || rp.seq >= 1) {  // (without GEBV: long rows only, cross.hip -- short ones keep the plain loop)
          if constexpr (TT == 1) {
            if (rp.dig_lds && rp.outb && !rp.obs_nt) {
              packed_stream_role<TT, MODE_CROSS, R, true, kShortRowVectors, false, 4, true>(
                  rp, block, wave, &rwl, reinterpret_cast<char *>(lds_raw) + (rp.dig_lds - 1), &dig_ready);
              return;
            }
          }
          if (!RIDE && rp.outb && rp.obs_nt) {
            if constexpr (!RIDE)  // (nothing rides on a launch whose observation wants non-temporal stores: the instance
                                  // that carries riders is the largest kernel of the library as it is)
              packed_stream_role<TT, MODE_CROSS, R, true, obs_vectors<TT>(), true>(rp, block, wave, &rwl);
          } else if (rp.outb)
            packed_stream_role<TT, MODE_CROSS, R, true, obs_vectors<TT>(), false>(rp, block, wave, &rwl);
          else
            packed_stream_role<TT, MODE_CROSS, R, false, TT == 0 ? kLongRowVectors : kNoObsVectors>(rp, block, wave, &rwl);
          return;
        }
      }
      packed_row_role<TT, MODE_CROSS, R, MW == 4 ? 0 : 3>(rp, block, wave, &rwl);
    }
  } else if constexpr (MW == 1) {
    if (block < mp.n_blocks) mask_role<1, SYNC_WAVE, LAYOUT>(mp, block, wave - 4, lds_raw, nullptr);
  } else {
    if (mp.sites) {  // long rows: one gamete stream per lane, site words to memory (mask_stream.h)
      mask_stream_role(mp, block * 4 + wave - 4, n_main * 4);
      return;
    }
    int iter = 0;  // crosses in turns when the grid is shorter than the mask work (cross_kernel)
    for (int mb = block; mb < mp.n_blocks; mb += n_main, ++iter)
      mask_role<4, SYNC_COUNTER, LAYOUT>(mp, mb, wave - 4, lds_raw, &counter, iter, &done);
  }
}

// Many long rows (HBM-resident): four row waves stream rows in turns without GEBV (packed_stream_role, TT = 0),
// MASK_WAVES waves draw the next generation's recombination sites one gamete stream per lane (mask_stream.h).
// No LDS, no barrier; the two kinds of wave never wait for each other.  More mask waves than row waves: the
// Threefry chains of a wave leave issue slots that only other waves can fill, and the workgroups of a grid of
// rows / (4 R seq) are few enough to be resident all at once (no tail of late workgroups doing mask work alone).
template <int R, int MASK_WAVES>
__global__ __launch_bounds__(64 * (4 + MASK_WAVES)) void packed_long_kernel(const PackedParams rp, const MaskParams mp) {
  prime_next_reward(rp.reward_next, rp.n_envs, rp.reward_pitch);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  static_assert(R == 1, "the LDS-DMA stream takes one row per turn");
  __shared__ __attribute__((aligned(16))) char ring[4][kDmaRingBytes];  // one ring per row wave (packed_dma.h)
  if (wave < 4) {
    if (rp.outb)
      packed_dma_role<MODE_CROSS, true>(rp, (int)blockIdx.x, wave, ring[wave]);
    else
      packed_dma_role<MODE_CROSS, false>(rp, (int)blockIdx.x, wave, ring[wave]);
  } else {
    mask_stream_role(mp, (int)blockIdx.x * MASK_WAVES + wave - 4, (int)gridDim.x * MASK_WAVES);
  }
}

// packed rows -> packed rows through an index, nothing else (no observation, no GEBV): what the population-sharded
// generation does with its winners before the all-gather (vector/distributed.py: a few hundred rows of 25 KB at
// configs[3]).  The row roles above give a whole row to one wave (196 dependent trips of 2-byte words per lane for a
// 100k-marker row: 67 us for 940 rows); here a wave takes 4 KB of a row, 16 bytes per lane, four loads in flight:
// grid (ceil(row bytes / 4096 / 4), rows).
struct CopyRowsParams {
  const uint8_t *src;  // (n_pool, row_bytes) packed rows
  uint8_t *dst;        // (n_rows, row_bytes)
  const int32_t *idx;  // (n_rows) source row of every output row (JAX retrieval indexing), or null (= identity)
  int64_t row_bytes;   // 16 * pw
  int32_t n_pool, n_rows;
};
__global__ __launch_bounds__(256) void packed_copy_rows_kernel(const CopyRowsParams p) {
  const int row = blockIdx.y, lane = threadIdx.x & 63;
  const int64_t piece0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 256 + lane;  // 16-byte pieces: 256 per wave
  const int64_t n_pieces = p.row_bytes >> 4;
  if (piece0 - lane >= n_pieces) return;
  const int i0 = p.idx ? norm_index(p.idx[row], p.n_pool) : row;
  const uint4 *s = reinterpret_cast<const uint4 *>(p.src + (int64_t)i0 * p.row_bytes);
  uint4 *d = reinterpret_cast<uint4 *>(p.dst + (int64_t)row * p.row_bytes);
  uint4 t[4];  // (every lane loads, one past the end its row's last piece: a load under a lane condition is a branch with a
               // wait for the load behind it -- four trips to memory in a row instead of four loads in flight)
#pragma unroll
  for (int u = 0; u < 4; ++u) t[u] = s[piece0 + 64 * u < n_pieces ? piece0 + 64 * u : n_pieces - 1];
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (piece0 + 64 * u < n_pieces) d[piece0 + 64 * u] = t[u];
}

// packed rows (through an index) -> byte-layout rows and nothing else: `unpack`, the observation of a lazy env.  A row
// cut over many waves like pack_kernel's: as a gather of the row roles -- a wave per row, 196 trips of 2-byte words for a
// 100k-marker row -- 2048 rows took 181 us where packing them takes 87.  grid (ceil(nvec / 2048), rows).
struct ExpandParams {
  const uint16_t *src;  // (n_pool, nvp)
  const int32_t *idx;   // (n_rows) or null (= identity)
  uint8_t *dst;         // (n_rows, pitch)
  int64_t pitch;
  int32_t nvec, nvp, n_pool, n_rows, row0, nt;
};
__global__ __launch_bounds__(256) void packed_expand_kernel(const ExpandParams p) {
  const int row = p.row0 + blockIdx.y, lane = threadIdx.x & 63;
  const int v0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 512 + lane;
  if (v0 - lane >= p.nvec) return;
  const uint16_t *s = p.src + (int64_t)(p.idx ? norm_index(p.idx[row], p.n_pool) : row) * p.nvp;
  char *d = reinterpret_cast<char *>(p.dst + (int64_t)row * p.pitch);
  uint32_t w[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) w[u] = s[min(v0 + 64 * u, p.nvec - 1)];  // (every lane loads)
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int v = v0 + 64 * u;
    if (v >= p.nvec) continue;
    const uint4 x = expand_vector(w[u]);
    if (p.nt)
      store_obs(d + 16 * (int64_t)v, x);
    else
      *reinterpret_cast<uint4 *>(d + 16 * (int64_t)v) = x;
  }
}

// byte-layout rows -> packed rows.  One lane per 16-byte vector.
struct PackParams {
  const uint8_t *src;  // (rows, pitch)
  uint16_t *dst;       // (rows, nvp)
  int64_t pitch;
  int32_t nvec, nvp, n_rows;
};

// grid (ceil(nvp / 2048), rows): a wave takes 512 consecutive vectors of one row, eight per lane 64 apart (every load a
// contiguous KB, all eight in flight before the first is used) -- a row is cut over many waves, so that also a few
// thousand long rows fill the chip (one wave per whole row read 416 MB in 135 us; this reads them in 85)
__global__ __launch_bounds__(256) void pack_kernel(const PackParams p) {
  const int row = blockIdx.y, lane = threadIdx.x & 63;
  const int v0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 512 + lane;
  if (v0 - lane >= p.nvp) return;
  const uint8_t *s = p.src + (int64_t)row * p.pitch;
  uint16_t *d = p.dst + (int64_t)row * p.nvp;
  uint4 w[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int v = v0 + 64 * u;
    // every lane loads, one past the end its row's last vector: under a lane condition -- also as "load, then select":
    // the compiler turns a select on a loaded value back into a branch -- the eight loads were eight trips to memory
    // in a row.  What a lane past the end packs is dropped below.
    w[u] = *reinterpret_cast<const uint4 *>(s + 16 * (int64_t)(v < p.nvec ? v : p.nvec - 1));
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int v = v0 + 64 * u;
    const uint32_t dw[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)  // bit 0 of bytes 0..3 -> bits 4k .. 4k+3
      o |= (((dw[k] & 0x01010101u) * 0x01020408u) >> 24) << (4 * k);
    if (v < p.nvp) d[v] = (uint16_t)(v < p.nvec ? o : 0u);  // zero words from the last marker up to the pitch
  }
}

}  // namespace bh
