// The mask role for long rows, one gamete stream per lane.
//
// mask_role (row_kernels.h) gives a LANE a marker pair and a WAVE one cross: per trip every lane needs its own
// thresholds -- vector loads.  Next to row waves that keep the memory pipeline full (100k-marker rows are
// HBM-resident) such a load waits its turn for microseconds, and four trips of look-ahead do not cover it: the
// role that is supposed to hide under the streaming ends up waiting for it (measured: without its table loads the
// two-role launch at 2k x 100k takes 590 us instead of 705, profiles/r02_exp_c3_roles_pmc_priorities.txt).
//
// Here a lane is a gamete STREAM -- (cross, gamete) -- and the wave walks the marker pairs in order, so that the
// pair index, both counters and both thresholds are wave-uniform: thresholds come through the scalar cache
// (s_load, a path of its own), the key schedule of a lane's stream sits in eight VGPRs, and the recombination
// sites of a lane are the carry chain of `w = w + w + (draw < threshold)`: one compare and one add-with-carry
// per draw.  No LDS, no ballots, no hand-off between waves, no vector load in the loop; per 64 pairs a lane
// stores its two site words (16 bytes) into the site array of its cross.
//
// What a wave cannot do this way is the prefix-XOR along the marker axis of a whole cross (its lanes hold 64
// different crosses): the site words go to memory and mask_finish_kernel -- phase B of mask_role with the site
// words read from there, one wave per cross -- turns them into crossover masks right behind the launch
// (2 x 51 MB at 2048 x 100k: ~30 us).
//
// Work = (group of 32 crosses [64 in double-haploid mode], trip of 64 pairs), group-major; every mask wave of the
// launch takes a contiguous share, so a wave derives stream keys once or twice.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "row_kernels.h"

namespace bh {

typedef const uint32_t __attribute__((address_space(4))) *scalar_u32_ptr;
typedef const uint64_t __attribute__((address_space(4))) *scalar_u64_ptr;

// w = 2 w + (draw < threshold): the site bits of a stream, newest in bit 0
#define BH_SITE(W_, DRAW_, THR_) \
  asm volatile("v_cmp_gt_u32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(W_) : "v"(DRAW_), "s"(THR_) : "vcc")

__device__ __forceinline__ void mask_stream_role(const MaskParams &p, const int wid, const int n_waves) {
  const int lane = threadIdx.x & 63;
  const int cpw = p.dh ? 64 : 32;  // crosses per group
  const int groups = (p.n_rows + cpw - 1) / cpw;
  const int total = groups * p.nwh;  // (group, trip) units, group-major
  const int per = (total + n_waves - 1) / n_waves;
  int u = __builtin_amdgcn_readfirstlane(wid * per);
  const int u_end = __builtin_amdgcn_readfirstlane(min(total, u + per));
  if (u >= u_end) return;
  if constexpr (kMaskPrio > 0) __builtin_amdgcn_s_setprio(kMaskPrio);

  const scalar_u32_ptr thrA = (scalar_u32_ptr)(uintptr_t)p.thrA, thrB = (scalar_u32_ptr)(uintptr_t)p.thrB;
  const uint32_t M = (uint32_t)p.M, half = (uint32_t)p.half;

  int cur_group = -1;
  uint32_t ks[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // this lane's key schedule: k1, k2+1, k2, k0+2, k0, k1+3, k2+4, k0+5
  int c = 0;
  bool valid = false;
  for (; u < u_end; ++u) {
    const int g = u / p.nwh, t = u - g * p.nwh;  // wave-uniform
    if (g != cur_group) {
      cur_group = g;
      // ---- stream keys: keys = split(sim_key, nkeys); gamete (c, q) owns keys[1 + 2c + q] (double haploid: the
      //      one gamete of c owns keys[1 + c]); its recombination key is split(gamete_key, 2)[0]
      const int cl = g * cpw + (p.dh ? lane : lane >> 1);
      valid = cl < p.n_rows;
      c = valid ? cl : p.n_rows - 1;
      const uint32_t q = p.dh ? 0u : (uint32_t)(lane & 1);
      const uint32_t cg = p.key_offset + (uint32_t)c;
      const uint32_t kidx = p.dh ? 1u + cg : 1u + 2u * cg + q;
      u32x2 rk;
      if (p.layout == LAYOUT_PARTITIONABLE) {
        const u32x2 gk = threefry2x32(p.key0, p.key1, 0u, kidx);
        rk = threefry2x32(gk.x, gk.y, 0u, 0u);
      } else {
        const u32x2 gk = split_at(p.key0, p.key1, p.nkeys, kidx);
        rk = split_at(gk.x, gk.y, 2u, 0u);
      }
      const KeySched k = key_sched(rk.x, rk.y);
#pragma unroll
      for (int i = 0; i < 8; ++i) ks[i] = k.w[i];
    }

    // ---- one trip: pairs j = 64 t .. 64 t + 63 in order.  Original layout: pair j draws marker j (word 0) and
    // marker j + half (word 1) from TF(key, (j, j + half)) [0 for the pad slot of an odd M]; two pairs run as
    // the two chains of the hand-placed block.  Partitionable layout: marker m draws o0 ^ o1 of TF(key, (0, m));
    // the two chains are markers j and j + half.
    uint32_t wA = 0, wB = 0, sAlo = 0, sBlo = 0;
    const uint32_t j0 = (uint32_t)t * 64u;
#pragma unroll 1
    for (uint32_t jb = j0; jb < j0 + 64u; jb += 8u) {
      uint32_t ta[8], tb[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ta[i] = thrA[jb + i];
        tb[i] = thrB[jb + i];
      }
      if (p.layout == LAYOUT_PARTITIONABLE) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const uint32_t j = jb + (uint32_t)i;
          const uint32_t c1 = j + half < M ? j + half : 0u;
          uint32_t a0 = ks[4], a1 = j + ks[0], b0 = ks[4], b1 = c1 + ks[0];
          asm volatile(BH_TFA BH_TFB
                       : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)
                       : "v"(ks[0]), "v"(ks[1]), "v"(ks[2]), "v"(ks[3]), "v"(ks[4]), "v"(ks[5]), "v"(ks[6]), "v"(ks[7]),
                         "v"(ks[0]), "v"(ks[1]), "v"(ks[2]), "v"(ks[3]), "v"(ks[4]), "v"(ks[5]), "v"(ks[6]), "v"(ks[7]));
          const uint32_t da = a0 ^ a1, db = b0 ^ b1;
          BH_SITE(wA, da, ta[i]);
          BH_SITE(wB, db, tb[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          const uint32_t j = jb + (uint32_t)i;
          const uint32_t c1a = j + half < M ? j + half : 0u, c1b = j + 1u + half < M ? j + 1u + half : 0u;
          uint32_t a0 = j + ks[4], a1 = c1a + ks[0], b0 = j + 1u + ks[4], b1 = c1b + ks[0];
          asm volatile(BH_TFA BH_TFB
                       : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)
                       : "v"(ks[0]), "v"(ks[1]), "v"(ks[2]), "v"(ks[3]), "v"(ks[4]), "v"(ks[5]), "v"(ks[6]), "v"(ks[7]),
                         "v"(ks[0]), "v"(ks[1]), "v"(ks[2]), "v"(ks[3]), "v"(ks[4]), "v"(ks[5]), "v"(ks[6]), "v"(ks[7]));
          BH_SITE(wA, a0, ta[i]);
          BH_SITE(wB, a1, tb[i]);
          BH_SITE(wA, b0, ta[i + 1]);
          BH_SITE(wB, b1, tb[i + 1]);
        }
      }
      if (jb == j0 + 24u) {  // 32 pairs done: bit 31 is the first of them
        sAlo = __builtin_bitreverse32(wA);
        sBlo = __builtin_bitreverse32(wB);
        wA = 0;
        wB = 0;
      }
    }
    uint64_t sA = ((uint64_t)__builtin_bitreverse32(wA) << 32) | sAlo;
    uint64_t sB = ((uint64_t)__builtin_bitreverse32(wB) << 32) | sBlo;
    if (p.alwA) {  // markers with r >= 1: always a site
      sA |= ((scalar_u64_ptr)(uintptr_t)p.alwA)[t];
      sB |= ((scalar_u64_ptr)(uintptr_t)p.alwB)[t];
    }
    if (valid) {
      // four planes per cross -- A and B words of gamete 0, of gamete 1 -- each spitch words long, so that
      // mask_finish_kernel reads consecutive trips from consecutive words
      uint64_t *Sc = p.sites + (int64_t)c * 4 * p.spitch + t;
      if (p.dh) {
        Sc[0] = sA;
        Sc[p.spitch] = sB;
        Sc[2 * (int64_t)p.spitch] = sA;
        Sc[3 * (int64_t)p.spitch] = sB;
      } else {
        Sc += 2 * (int64_t)(lane & 1) * p.spitch;
        Sc[0] = sA;
        Sc[p.spitch] = sB;
      }
    }
  }
}

#undef BH_SITE

// ---- rows beyond what one workgroup's LDS holds (more than 2^18 markers) -------------------------------------
// mask_role keeps a cross's site words in LDS between its two phases: M / 4 bytes, 64 KB at 2^18 markers.  Longer
// rows (the reference's own cross timing uses 1,406,757 markers, scripts/time_cross.py:14) take the two phases as
// two kernels with the site words in memory, like the stream-per-lane role above -- but with mask_role's mapping,
// a LANE per marker pair and any number of waves per cross, because such calls have few crosses (hundreds, not
// thousands: 64 gamete streams per wave would leave most of the chip idle).  The marker axis is cut into tiles of
// 64 trips (4096 pairs) per workgroup; the prefix-XOR that runs along the whole row is mask_finish_kernel's, whose
// carry between its 64-word blocks is the one bit per gamete that has to cross a tile boundary.
constexpr int kSiteTileTrips = 64;  // trips of 64 marker pairs per workgroup (16 per wave)

__global__ __launch_bounds__(256) void mask_sites_kernel(const MaskParams p) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tiles = (p.nwh + kSiteTileTrips - 1) / kSiteTileTrips;
  const int c = (int)blockIdx.x / tiles, tile = (int)blockIdx.x - c * tiles;
  if (c >= p.n_rows) return;
  // gamete keys as in mask_role: lanes 0..3 carry the four words of the two recombination keys
  uint32_t rk00, rk01, rk10, rk11;
  {
    const uint32_t q = (lane >> 1) & 1u, w = lane & 1u;
    const uint32_t cg = p.key_offset + (uint32_t)c;
    const uint32_t kidx = p.dh ? 1u + cg : 1u + 2u * cg + q;
    uint32_t rw;
    if (p.layout == LAYOUT_PARTITIONABLE) {
      const u32x2 gk = threefry2x32(p.key0, p.key1, 0u, kidx);
      const u32x2 rk = threefry2x32(gk.x, gk.y, 0u, 0u);
      rw = w ? rk.y : rk.x;
    } else {
      const uint32_t gw = iota_word_even(p.key0, p.key1, 2u * p.nkeys, 2u * kidx + w);
      const uint32_t g0 = __shfl(gw, lane & ~1), g1 = __shfl(gw, lane | 1);
      rw = threefry2x32(g0, g1, w, w + 2u).x;
    }
    rk00 = __builtin_amdgcn_readlane(rw, 0);
    rk01 = __builtin_amdgcn_readlane(rw, 1);
    rk10 = __builtin_amdgcn_readlane(rw, 2);
    rk11 = __builtin_amdgcn_readlane(rw, 3);
  }
  const KeySched ka = key_sched(rk00, rk01), kb = key_sched(rk10, rk11);
  uint64_t *S = p.sites + (int64_t)c * 4 * p.spitch;
  const int t_end = min(p.nwh, (tile + 1) * kSiteTileTrips);
  for (int t = tile * kSiteTileTrips + wv; t < t_end; t += 4) {
    const uint32_t j = (uint32_t)t * 64u + (uint32_t)lane;
    const uint32_t tA = p.thrA[j], tB = p.thrB[j], c1 = p.ctrB[j];  // tables are padded past the last pair
    uint64_t sA0, sB0, sA1, sB1;
    if (p.layout == LAYOUT_PARTITIONABLE) {  // marker m draws o0 ^ o1 of TF(key, (0, m)): pairs [0, half), then [half, M)
      uint32_t a0 = rk00, a1 = j + rk01, b0 = rk10, b1 = j + rk11;
      asm volatile(BH_TFA BH_TFB
                   : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)
                   : "s"(ka.w[0]), "s"(ka.w[1]), "s"(ka.w[2]), "s"(ka.w[3]), "s"(ka.w[4]), "s"(ka.w[5]), "s"(ka.w[6]),
                     "s"(ka.w[7]), "s"(kb.w[0]), "s"(kb.w[1]), "s"(kb.w[2]), "s"(kb.w[3]), "s"(kb.w[4]), "s"(kb.w[5]),
                     "s"(kb.w[6]), "s"(kb.w[7]));
      uint32_t c0 = rk00, c1v = c1 + rk01, d0 = rk10, d1 = c1 + rk11;
      asm volatile(BH_TFA BH_TFB
                   : "+v"(c0), "+v"(c1v), "+v"(d0), "+v"(d1)
                   : "s"(ka.w[0]), "s"(ka.w[1]), "s"(ka.w[2]), "s"(ka.w[3]), "s"(ka.w[4]), "s"(ka.w[5]), "s"(ka.w[6]),
                     "s"(ka.w[7]), "s"(kb.w[0]), "s"(kb.w[1]), "s"(kb.w[2]), "s"(kb.w[3]), "s"(kb.w[4]), "s"(kb.w[5]),
                     "s"(kb.w[6]), "s"(kb.w[7]));
      sA0 = __ballot((a0 ^ a1) < tA);
      sB0 = __ballot((c0 ^ c1v) < tB);
      sA1 = __ballot((b0 ^ b1) < tA);
      sB1 = __ballot((d0 ^ d1) < tB);
    } else {  // pair j draws marker j (word 0) and marker j + half (word 1) of TF(key, (j, j + half))
      uint32_t a0 = j + rk00, a1 = c1 + rk01, b0 = j + rk10, b1 = c1 + rk11;
      asm volatile(BH_TFA BH_TFB
                   : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)
                   : "s"(ka.w[0]), "s"(ka.w[1]), "s"(ka.w[2]), "s"(ka.w[3]), "s"(ka.w[4]), "s"(ka.w[5]), "s"(ka.w[6]),
                     "s"(ka.w[7]), "s"(kb.w[0]), "s"(kb.w[1]), "s"(kb.w[2]), "s"(kb.w[3]), "s"(kb.w[4]), "s"(kb.w[5]),
                     "s"(kb.w[6]), "s"(kb.w[7]));
      sA0 = __ballot(a0 < tA);
      sB0 = __ballot(a1 < tB);
      sA1 = __ballot(b0 < tA);
      sB1 = __ballot(b1 < tB);
    }
    if (p.alwA) {  // markers with r >= 1: always a site
      const uint64_t xa = p.alwA[t], xb = p.alwB[t];
      sA0 |= xa;
      sA1 |= xa;
      sB0 |= xb;
      sB1 |= xb;
    }
    if (lane == 0) {  // the planes mask_finish_kernel reads: A and B words of gamete 0, then of gamete 1
      S[t] = sA0;
      S[p.spitch + t] = sB0;
      S[2 * (int64_t)p.spitch + t] = sA1;
      S[3 * (int64_t)p.spitch + t] = sB1;
    }
  }
}

// site_word (row_kernels.h) for the planar site array of one cross: the 64 site bits of markers [64w, 64w + 64) of
// gamete g; plane 2g holds the A words (markers [0, half) from bit 0 of word 0), plane 2g + 1 the B words
// (markers [half, M)); the word behind the last one of a plane is zero.
// Without a branch, for any w in [0, nw): a word of the A plane (or its zero word), and 16 bytes of the B plane's string
// from the word the bit 64 w - half lies in -- the word in FRONT of plane B is plane A's zero word, the word behind its
// last its own.  (Round 5: as three cases under lane conditions -- whole word in A, whole word in B at a bit offset
// from two loads, the one word that straddles -- the finish kernels came out as 190 branches with a wait for the loads
// behind each: "all loads in flight together" was not what ran.  2048 x 100k: 37 -> 29 us, profiles/r05_ab_finish.txt.)
__device__ __forceinline__ uint64_t site_word_planar(const uint64_t *S, int spitch, int g, int w, int half) {
  const uint64_t *SA = S + (int64_t)(2 * g) * spitch, *SB = SA + spitch;
  const int nwh = spitch - 1;
  const uint64_t a = SA[w < nwh ? w : nwh];
  const int off = (w << 6) - half;  // the word's first bit in plane B's string (negative: in front of it)
  const int k = (off >> 6) > -1 ? (off >> 6) : -1, sh = off & 63;
  typedef uint64_t u64x2 __attribute__((ext_vector_type(2), aligned(8)));
  const u64x2 b = *reinterpret_cast<const u64x2 *>(SB + k);
  const uint64_t lo = b.x >> sh;
  const uint64_t hi = (sh != 0 && off >= -64) ? b.y << (64 - sh) : 0ull;
  return a | lo | hi;
}

// Phase B of mask_role over site words in memory: one wave per cross, inclusive prefix-XOR of the site strings,
// masks written bit-interleaved (see mask_role).
__global__ __launch_bounds__(256) void mask_finish_kernel(const MaskParams p) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (c >= p.n_rows) return;
  const uint64_t *S = p.sites + (int64_t)c * 4 * p.spitch;
  uint8_t *Z = p.Z + (int64_t)c * p.zpitch;
  uint32_t carry0 = 0, carry1 = 0;
  const uint64_t below = (1ull << lane) - 1ull;
  constexpr int U = 4;  // blocks of 64 words whose site words are requested together: the loop is one round trip to
                        // memory per turn, and a cross has only nw / 64 of them (25 at 100k markers)
  for (int w0 = 0; w0 < p.nw; w0 += 64 * U) {
    uint64_t s0[U], s1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int w = w0 + 64 * u + lane;
      const int wc = w < p.nw ? w : p.nw - 1;  // (every lane loads: a load under a lane condition is a branch)
      const uint64_t v0 = site_word_planar(S, p.spitch, 0, wc, p.half), v1 = site_word_planar(S, p.spitch, 1, wc, p.half);
      s0[u] = w < p.nw ? v0 : 0ull;
      s1[u] = w < p.nw ? v1 : 0ull;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int w = w0 + 64 * u + lane;
      uint64_t x0 = prefix_xor64(s0[u]), x1 = prefix_xor64(s1[u]);
      const uint64_t par0 = __ballot((x0 >> 63) != 0), par1 = __ballot((x1 >> 63) != 0);
      if ((__popcll(par0 & below) & 1u) ^ carry0) x0 = ~x0;
      if ((__popcll(par1 & below) & 1u) ^ carry1) x1 = ~x1;
      carry0 ^= __popcll(par0) & 1u;
      carry1 ^= __popcll(par1) & 1u;
      if (w < p.nw) {
        uint4 z;
        z.x = spread16((uint32_t)x0 & 0xffffu) | (spread16((uint32_t)x1 & 0xffffu) << 1);
        z.y = spread16((uint32_t)x0 >> 16) | (spread16((uint32_t)x1 >> 16) << 1);
        z.z = spread16((uint32_t)(x0 >> 32) & 0xffffu) | (spread16((uint32_t)(x1 >> 32) & 0xffffu) << 1);
        z.w = spread16((uint32_t)(x0 >> 48)) | (spread16((uint32_t)(x1 >> 48)) << 1);
        *reinterpret_cast<uint4 *>(Z + 16 * (int64_t)w) = z;
      }
    }
  }
}

// The same phase B with FOUR waves per cross, for launches with few keyed crosses (BASELINE configs[2]: 2048): one
// wave per cross walks a 100k-marker row in seven dependent round trips to memory (43 us for 2048 crosses: latency, not
// bytes).  Here every wave takes a quarter of the row: it first reads its quarter once for nothing but the parity of
// its site bits (all loads in flight together), the four parities meet in LDS -- the crossover state a quarter starts
// in is the XOR of the quarters before it -- and then it runs the prefix-XOR over its quarter from that state (the
// second read comes out of the L2).  Two round trips instead of seven, four times the waves.
__global__ __launch_bounds__(256) void mask_finish4_kernel(const MaskParams p) {
  __shared__ uint32_t par[4][2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x;  // grid = n_rows
  const uint64_t *S = p.sites + (int64_t)c * 4 * p.spitch;
  uint8_t *Z = p.Z + (int64_t)c * p.zpitch;
  const int quarter = ((p.nw + 255) / 256) * 64;  // words per wave, a multiple of 64
  const int w_lo = wv * quarter, w_hi = min(p.nw, w_lo + quarter);
  // up to 512 words per quarter (131k markers) the site words stay in registers between the two passes
  constexpr int KEEP = 8;
  const bool keep = quarter <= 64 * KEEP;  // uniform
  uint64_t k0[KEEP], k1[KEEP];
  {
    uint32_t q0 = 0, q1 = 0;
    if (keep) {
#pragma unroll
      for (int u = 0; u < KEEP; ++u) {
        const int w = w_lo + 64 * u + lane;
        const int wc = w < p.nw ? w : p.nw - 1;  // (every lane loads: a load under a lane condition is a branch)
        const uint64_t v0 = site_word_planar(S, p.spitch, 0, wc, p.half), v1 = site_word_planar(S, p.spitch, 1, wc, p.half);
        k0[u] = w < w_hi ? v0 : 0ull;
        k1[u] = w < w_hi ? v1 : 0ull;
      }
#pragma unroll
      for (int u = 0; u < KEEP; ++u) {
        q0 ^= (uint32_t)__popcll(k0[u]);
        q1 ^= (uint32_t)__popcll(k1[u]);
      }
    } else {
      for (int w = w_lo + lane; w < w_hi; w += 64) {
        q0 ^= (uint32_t)__popcll(site_word_planar(S, p.spitch, 0, w, p.half));
        q1 ^= (uint32_t)__popcll(site_word_planar(S, p.spitch, 1, w, p.half));
      }
    }
    const uint32_t t0 = (uint32_t)__popcll(__ballot(q0 & 1u)) & 1u, t1 = (uint32_t)__popcll(__ballot(q1 & 1u)) & 1u;
    if (lane == 0) {
      par[wv][0] = t0;
      par[wv][1] = t1;
    }
  }
  __syncthreads();
  uint32_t carry0 = 0, carry1 = 0;
  for (int k = 0; k < wv; ++k) {
    carry0 ^= par[k][0];
    carry1 ^= par[k][1];
  }
  const uint64_t below = (1ull << lane) - 1ull;
  auto block = [&](int w, uint64_t s0, uint64_t s1) {  // one 64-word block: prefix-XOR, then the interleaved mask words
    uint64_t x0 = prefix_xor64(s0), x1 = prefix_xor64(s1);
    const uint64_t par0 = __ballot((x0 >> 63) != 0), par1 = __ballot((x1 >> 63) != 0);
    if ((__popcll(par0 & below) & 1u) ^ carry0) x0 = ~x0;
    if ((__popcll(par1 & below) & 1u) ^ carry1) x1 = ~x1;
    carry0 ^= __popcll(par0) & 1u;
    carry1 ^= __popcll(par1) & 1u;
    if (w < w_hi) {
      uint4 z;
      z.x = spread16((uint32_t)x0 & 0xffffu) | (spread16((uint32_t)x1 & 0xffffu) << 1);
      z.y = spread16((uint32_t)x0 >> 16) | (spread16((uint32_t)x1 >> 16) << 1);
      z.z = spread16((uint32_t)(x0 >> 32) & 0xffffu) | (spread16((uint32_t)(x1 >> 32) & 0xffffu) << 1);
      z.w = spread16((uint32_t)(x0 >> 48)) | (spread16((uint32_t)(x1 >> 48)) << 1);
      *reinterpret_cast<uint4 *>(Z + 16 * (int64_t)w) = z;
    }
  };
  if (keep) {
#pragma unroll
    for (int u = 0; u < KEEP; ++u)
      if (w_lo + 64 * u < w_hi) block(w_lo + 64 * u + lane, k0[u], k1[u]);  // uniform
  } else {
    constexpr int U = 4;
    for (int w0 = w_lo; w0 < w_hi; w0 += 64 * U) {
      uint64_t s0[U], s1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int w = w0 + 64 * u + lane;
        const int wc = w < p.nw ? w : p.nw - 1;
        const uint64_t v0 = site_word_planar(S, p.spitch, 0, wc, p.half), v1 = site_word_planar(S, p.spitch, 1, wc, p.half);
        s0[u] = w < w_hi ? v0 : 0ull;
        s1[u] = w < w_hi ? v1 : 0ull;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) block(w0 + 64 * u + lane, s0[u], s1[u]);
    }
  }
}

}  // namespace bh
