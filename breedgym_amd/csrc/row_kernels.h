// Device kernels for the BreedGym hot path on gfx950 (MI355X, CDNA4, wave64).
//
// A cross is done by two kernels that meet in a small mask buffer:
//
//   mask_kernel  (VALU-bound)  Threefry-2x32 per marker pair (j, j+half) for both gametes of a
//     cross; the Bernoulli(r) recombination sites leave the wave as ballots -> 64-bit site
//     words in LDS (phase A); an inclusive prefix-XOR over the site strings (6 shift-xor steps
//     inside a 64-bit word, ballot+popcount across words) turns them into crossover masks,
//     written to HBM bit-interleaved, 2 bits per marker = M/4 bytes per cross (phase B).
//     One Threefry call serves two markers (JAX's original random_bits layout hashes the
//     counter vector in two halves) and each threshold load is shared by the two gametes.
//     The masks depend on the key chain and the cross slot only -- not on the parents -- so
//     they are computed one generation AHEAD: cross_kernel is a single launch whose workgroups
//     take one of two roles, "row" (stream generation g with the masks made by the previous
//     launch) or "mask" (make the masks of generation g+1).  The two kinds of workgroup share
//     every CU, so the VALU-bound Threefry work runs under the memory-bound streaming.
//   row_kernel   (HBM-bound)   16-byte coalesced reads of both parents through the action
//     indices (8 markers x 2 alleles per lane; the reference's materialised parent gather,
//     vec_env.py:89-90, is just an address here), one v_perm_b32 per output dword picks the
//     allele the mask bits point at, 16-byte coalesced store of the offspring, and the GEBV
//     dot product is accumulated in f64 from the bytes already in registers.  The same kernel
//     in MODE_GATHER / MODE_GEBV is the founder gather of reset() and the standalone GEBV.
//
// A group of WAVES wavefronts owns one row.  Workgroups are always 256 threads = 4 waves, one
// per SIMD of the CU: with WAVES = 1 a workgroup carries four independent rows, with
// WAVES = 4 (long rows) the whole workgroup shares one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "threefry.h"

constexpr int kMaskPrio = 1;  // priority of mask waves next to row waves

namespace bh {

enum { MODE_CROSS = 0, MODE_GATHER = 1, MODE_GEBV = 2 };

struct MaskParams {
  uint8_t *Z;            // (n_rows, zpitch) crossover masks, the two gametes bit-interleaved, out
  const uint32_t *thrA;  // [(nwh+16)*64] 32-bit thresholds of markers [0, half), zero padded
  const uint32_t *thrB;  // [(nwh+16)*64] 32-bit thresholds of markers [half, M), zero padded
  const uint32_t *ctrB;  // [(nwh+16)*64] second Threefry counter of pair j: j + half, or 0 where that is >= M
  const uint64_t *alwA;  // [nwh] site words that are always set (r >= 1), or null if there are none
  const uint64_t *alwB;
  int64_t zpitch;        // bytes per cross = 16 * pw (ctx.h): words nw .. pw - 1 are never written
  int32_t M, half, nwh, nw;
  int32_t n_rows;     // B: crosses with their own keys
  int32_t n_blocks;   // workgroups with mask waves = ceil(n_rows / (4 / WAVES))
  int32_t group_lds;  // LDS bytes per group (multiple of 16)
  uint32_t thr_bytes; // size of thrA / thrB in bytes (buffer descriptor range)
  uint32_t key0, key1;  // the call's simulator key
  uint32_t nkeys;       // keys split off it: 2*B + 1, or B + 1 for double haploids
  int32_t dh;           // 1: both gametes of cross c use keys[1 + c] (one gamete, doubled)
  int32_t layout;       // LAYOUT_ORIGINAL / LAYOUT_PARTITIONABLE (threefry.h)
  uint64_t *sites;      // mask_stream_role / mask_finish_kernel (mask_stream.h): (n_rows, spitch, 4) site words in
  int32_t spitch;       // memory, spitch = nwh + 1 trips per cross (the last one stays zero); null: mask_role
  uint32_t *err;        // device-visible error word (host memory): a hand-off wait that ran out sets a bit
  uint32_t key_offset;  // cross c of this call is cross key_offset + c of the whole (sharded) batch the keys are
                        // split for: nkeys counts that whole batch (bh_set_cross_window)
};

__device__ __forceinline__ int xcd_row(int b, int n);

struct RowParams {
  const uint8_t *src;  // cross: (E, n_pool, pitch) pool; gather: (n_pool, pitch); gebv: (rows, pitch)
  uint8_t *out;        // (rows, pitch) or null (gebv)
  const int32_t *idx;  // cross: (rows, 2); gather: (rows); gebv: null
  float *gebv;         // (rows, T) or null
  const uint8_t *Z;    // cross: (B, zpitch) masks from mask_kernel
  const double *eff;   // (tcount, effPitch) f64 marker effects of traits t0.., zero padded
  const uint4 *dig;    // (T, 3, nvec + 1) int16 digit planes of the fixed-point effects (row_role_pipelined)
  const double *scale; // (T) 2^-s_t
  int64_t pitch;       // bytes per genotype row (multiple of 16)
  int64_t effPitch;    // doubles per trait (= 8 * nvec)
  int64_t zpitch;
  int32_t nvec;
  int32_t n_pool, n_out;  // rows per env in src / out (cross); n_pool = src rows (gather)
  int32_t n_rows;         // total output rows
  int32_t n_blocks;       // workgroups with row waves = ceil(n_rows / (R * 4 / WAVES))
  int32_t T, t0, tcount;  // GEBV row stride, first trait and trait count of this launch
  int32_t rng_shared;     // 1: mask row = env-local cross index (all envs share the masks)
  float *reward;          // cross: (n_envs) max GEBV per env, holding -inf on entry (reward_note), or null
  float *reward_next;     // (n_envs) set to -inf by this launch for a later one, or null
  int64_t reward_pitch;   // floats from one env's word to the next in both (64: a cache line of its own per env --
                          // the atomics of different envs then do not queue behind each other: +0.9 us instead of
                          // +2.6 us on the 16 x 256 x 10k launch)
  int32_t n_envs;
};

// a reward buffer that was not left primed by a previous launch
__global__ void prime_reward_kernel(float *reward, int n_envs, int64_t pitch) {
  for (int64_t i = threadIdx.x; i < n_envs; i += blockDim.x) reward[i * pitch] = -INFINITY;
}

// reward[env] = max(reward[env], v) for a buffer that held -inf before the launch: the sign decides between
// a signed max and an unsigned min on the same bits (non-negative floats order like ints, negative ones in
// reverse like uints).  Device scope, no return value, no ordering needed: max commutes.
__device__ __forceinline__ void reward_note(float *reward, int env, float v) {
  if (v != v) return;
  const unsigned u = __float_as_uint(v);
  if (u >> 31)
    atomicMin(reinterpret_cast<unsigned *>(reward) + env, u);
  else
    atomicMax(reinterpret_cast<int *>(reward) + env, (int)u);
}

// Same-address device-scope atomics are served one after the other at the memory side (~170 ns each: 256
// per env and launch cost 40 us at 16 x 256 crosses), so the row waves of a workgroup first meet in LDS --
// maxima of order-preserving keys, then an arrival count -- and the last one notes the workgroup's value:
// one global atomic per workgroup and env.  A workgroup's rows lie in at most two envs whenever an env has
// at least as many rows as the workgroup; a row of a third env goes to memory directly.
struct RewardLds {
  unsigned count;
  unsigned best[2];  // order-preserving keys of the maxima of the workgroup's first and second env; 0 = none
};
__device__ __forceinline__ unsigned reward_key(float f) {
  const unsigned u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float reward_unkey(unsigned k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
__device__ __forceinline__ void reward_lds_init(RewardLds *rw) {  // thread 0, before a workgroup barrier
  rw->count = 0;
  rw->best[0] = 0;
  rw->best[1] = 0;
}
// one lane per wave: `row`'s value
__device__ __forceinline__ void reward_row(RewardLds *rw, float *reward, int64_t pitch, int env0, int row, int n_out,
                                           float v) {
  if (v != v) return;
  const int env = row / n_out;
  if (rw && env - env0 < 2)
    atomicMax(&rw->best[env - env0], reward_key(v));
  else
    reward_note(reward + (int64_t)env * pitch, 0, v);
}
// one lane per wave, after its reward_row calls; `expected` = row waves of this workgroup that have rows
__device__ __forceinline__ void reward_wave_done(RewardLds *rw, float *reward, int64_t pitch, int env0, int expected) {
  if (!rw) return;
  if (atomicAdd(&rw->count, 1u) + 1u == (unsigned)expected) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const unsigned k = atomicMax(&rw->best[s], 0u);
      if (k) reward_note(reward + (int64_t)(env0 + s) * pitch, 0, reward_unkey(k));
    }
  }
}
// first row of a workgroup of four row waves with `per_wave` rows each, the env it lies in, and how many of
// its waves have rows
__device__ __forceinline__ void reward_wg_layout(int block, int n_blocks, int per_wave, int n_rows, int n_out,
                                                 int &env0, int &expected) {
  const int first = xcd_row(block, n_blocks) * 4 * per_wave;
  env0 = first / n_out;
  const int left = n_rows - first;
  expected = left <= 0 ? 0 : (left >= 4 * per_wave ? 4 : (left + per_wave - 1) / per_wave);
}

// what the first workgroup of every launch does for the launch after it
__device__ __forceinline__ void prime_next_reward(float *reward_next, int n_envs, int64_t pitch, bool first);
__device__ __forceinline__ void prime_next_reward(float *reward_next, int n_envs, int64_t pitch) {
  prime_next_reward(reward_next, n_envs, pitch, blockIdx.x == 0);
}
__device__ __forceinline__ void prime_next_reward(float *reward_next, int n_envs, int64_t pitch, bool first) {
  if (reward_next && first)
    for (int64_t i = threadIdx.x; i < n_envs; i += blockDim.x) reward_next[i * pitch] = -INFINITY;
}

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

// 16 bits -> 32 bits with a zero after every bit
__device__ __forceinline__ uint32_t spread16(uint32_t x) {
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
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

// Site words live in LDS as S[t][4] = {A(g0), B(g0), A(g1), B(g1)}: A words hold markers
// [0, half) from bit 0 of word 0, B words markers [half, M) from bit 0 of word 0.
// Returns the 64 site bits of markers [64w, 64w+64) of gamete g.
__device__ __forceinline__ uint64_t site_word(const uint64_t *S, int g, int w, int half) {
  const uint64_t *SA = S + 2 * g, *SB = S + 2 * g + 1;
  const int base = w << 6;
  if (base + 64 <= half) return SA[4 * w];
  if (base >= half) {
    const int off = base - half, k = off >> 6, sh = off & 63;
    const uint64_t lo = SB[4 * k];
    return sh ? ((lo >> sh) | (SB[4 * (k + 1)] << (64 - sh))) : lo;
  }
  return SA[4 * w] | (SB[0] << (half - base));
}

// How the WAVES waves of a mask group meet between phase A and phase B.
enum { SYNC_WAVE = 0, SYNC_BLOCK = 1, SYNC_COUNTER = 2 };

// (the hand-placed Threefry block BH_TFA / BH_TFB and its key schedule live in threefry.h)

// `wave` = this wave's index among the four waves that play the role in this workgroup.
// SYNC says how the group's waves hand phase A's site words to the wave that runs phase B:
//   SYNC_WAVE    one wave per cross: program order (LDS operations of a wave execute in order)
//   SYNC_BLOCK   the workgroup is the group: __syncthreads()
//   SYNC_COUNTER four mask waves next to four row waves that must not be held up: every mask
//                wave bumps an LDS counter when its site words are written, wave 0 waits for it.
//                `counter` was zeroed before a workgroup barrier at kernel entry.
// SYNC_COUNTER groups may be called for several crosses in turn (`iter` = 0, 1, ..: the workgroup's
// share of the mask work, see cross_kernel): `counter` then counts arrivals over all turns, and
// `done` (the number of finished prefix passes) keeps the other waves from overwriting site words
// wave 0 is still reading.
// LAYOUT: the Threefry counter layout as a compile-time fact (LAYOUT_ORIGINAL / LAYOUT_PARTITIONABLE) where the caller
// is an instance of its own per layout -- the two-role launches of bit-plane rows, whose kernels would otherwise carry
// both layouts' hand-placed loops and grow past what a branch reaches (packed_kernels.h) -- or -1: p.layout decides.
template <int WAVES, int SYNC, int LAYOUT = -1>
__device__ __forceinline__ void mask_role(const MaskParams &p, const int block, const int wave, uint8_t *lds_raw,
                                          unsigned *counter, const int iter = 0, unsigned *done = nullptr) {
  const bool partitionable = LAYOUT < 0 ? p.layout == LAYOUT_PARTITIONABLE : LAYOUT == LAYOUT_PARTITIONABLE;
  static_assert(WAVES == 1 || WAVES == 4, "a group is one wave or four");
  static_assert((WAVES == 1) == (SYNC == SYNC_WAVE), "one wave needs no hand-off, four need one");
  constexpr int GROUPS = 4 / WAVES;
  const int lane = threadIdx.x & 63;
  const int grp = wave / WAVES, wv = wave % WAVES;
  const int tid = wv * 64 + lane;
  const int c = __builtin_amdgcn_readfirstlane(block * GROUPS + grp);
  if (c >= p.n_rows) return;  // only possible for WAVES == 1, whose waves never meet at a barrier
  // four-wave groups run next to row waves at the default priority (few keyed crosses, the launch
  // is bound by the streaming): one step above them the mask role is out of the way sooner --
  // 1-5 % on most shapes, 11 % at 2200 x 16k markers, -2 % on one (profiles/r01_ab_mask_priority.txt)
  if constexpr (SYNC == SYNC_COUNTER && kMaskPrio > 0) __builtin_amdgcn_s_setprio(kMaskPrio);
  uint64_t *S = reinterpret_cast<uint64_t *>(lds_raw + (size_t)grp * p.group_lds);  // [nwh + 1][4]

  // ---- gamete keys: keys = split(sim_key, nkeys); gamete (c, q) owns keys[1 + 2c + q] (double
  //      haploid: the one gamete of c owns keys[1 + c] and fills both slots); its
  //      recombination key is split(gamete_key, 2)[0].  Lanes 0..3 carry the 4 words.
  uint32_t rk00, rk01, rk10, rk11;
  {
    const uint32_t q = (lane >> 1) & 1u, w = lane & 1u;
    const uint32_t cg = p.key_offset + (uint32_t)c;
    const uint32_t kidx = p.dh ? 1u + cg : 1u + 2u * cg + q;
    uint32_t rw;
    if (partitionable) {  // key k = TF(key, (0, k)); lanes 2q, 2q+1 both make gamete q's
      const u32x2 gk = threefry2x32(p.key0, p.key1, 0u, kidx);
      const u32x2 rk = threefry2x32(gk.x, gk.y, 0u, 0u);  // split(gamete_key, 2)[0]
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
  const bool one_gamete = p.dh != 0;  // wave-uniform
  if constexpr (SYNC == SYNC_COUNTER) {
    if (iter > 0) {  // the previous cross's prefix pass must be through with the site words
      bool ok = false;
      for (int spin = 0; spin < (1 << 24); ++spin) {
        if (__hip_atomic_load(done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= (unsigned)iter) {
          ok = true;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      // all waves of a workgroup are resident, so the wait is finite; should it run out all the same, the
      // masks are not to be trusted: say so where the host looks (check_ctx, bh_sync)
      if (!ok && lane == 0 && p.err) __hip_atomic_fetch_or(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }

  // ---- phase A: recombination sites.  Everything around the Threefry block is kept off the
  // vector ALU where possible: thresholds come through a buffer descriptor (lane offset in a
  // VGPR, trip offset in an SGPR: no address arithmetic), one trip ahead of their use; the
  // compare is on the raw 32-bit word (thresholds pre-shifted); markers with r >= 1 are OR-ed in
  // as whole scalar words; the four ballots of a trip go to LDS with two 16-byte stores.
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void *)p.thrA, 0, p.thr_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void *)p.thrB, 0, p.thr_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void *)p.ctrB, 0, p.thr_bytes, 0x00020000);
  const uint32_t voff = lane * 4u;

  /* the second counter comes from a table too (j + half, but 0 for the pad slot of an odd-length    \
     stream): no compare/select in the loop */                                                                     \
#define BH_TRIP(T_, J_, TA_, TB_, C1_)                                                                              \
  {                                                                                                                 \
    uint32_t a0 = (J_) + rk00, a1 = (C1_) + rk01, b0 = (J_) + rk10, b1 = (C1_) + rk11;                              \
    uint64_t sA0, sB0, sA1, sB1;                                                                                    \
    if (one_gamete) { /* doubled haploids: both slots are the ONE gamete of keys[1 + c] -- one chain, not two */     \
      asm volatile(BH_TFA                                                                                           \
                   : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)                                                         \
                   : "s"(ka.w[0]), "s"(ka.w[1]), "s"(ka.w[2]), "s"(ka.w[3]), "s"(ka.w[4]), "s"(ka.w[5]),            \
                     "s"(ka.w[6]), "s"(ka.w[7]));                                                                   \
      sA0 = sA1 = __ballot(a0 < (TA_));                                                                             \
      sB0 = sB1 = __ballot(a1 < (TB_));                                                                             \
    } else {                                                                                                        \
      asm volatile(BH_TFA BH_TFB                                                                                    \
                   : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)                                                         \
                   : "s"(ka.w[0]), "s"(ka.w[1]), "s"(ka.w[2]), "s"(ka.w[3]), "s"(ka.w[4]), "s"(ka.w[5]),            \
                     "s"(ka.w[6]), "s"(ka.w[7]), "s"(kb.w[0]), "s"(kb.w[1]), "s"(kb.w[2]), "s"(kb.w[3]),            \
                     "s"(kb.w[4]), "s"(kb.w[5]), "s"(kb.w[6]), "s"(kb.w[7]));                                       \
      sA0 = __ballot(a0 < (TA_));                                                                                   \
      sB0 = __ballot(a1 < (TB_));                                                                                   \
      sA1 = __ballot(b0 < (TA_));                                                                                   \
      sB1 = __ballot(b1 < (TB_));                                                                                   \
    }                                                                                                               \
    if (p.alwA) {                                                                                                   \
      const uint64_t xa = p.alwA[(T_)], xb = p.alwB[(T_)];                                                          \
      sA0 |= xa;                                                                                                    \
      sA1 |= xa;                                                                                                    \
      sB0 |= xb;                                                                                                    \
      sB1 |= xb;                                                                                                    \
    }                                                                                                               \
    if (lane == 0) {                                                                                                \
      ulonglong2 *dstw = reinterpret_cast<ulonglong2 *>(S + 4 * (T_));                                              \
      dstw[0] = make_ulonglong2(sA0, sB0);                                                                          \
      dstw[1] = make_ulonglong2(sA1, sB1);                                                                          \
    }                                                                                                               \
  }

  /* The partitionable layout draws every marker from its own call, counter (0, m), word o0 ^ o1: the same    \
     two-chain block once for the markers [0, half) of both gametes and once for [half, M) (their counters    \
     are the second-counter table of the original layout; its 0 for the slot past an odd M meets a zero        \
     threshold).  Twice the integer work per marker. */                                                         \
#define BH_TRIP_P(T_, J_, TA_, TB_, C1_)                                                                            \
  {                                                                                                                 \
    uint32_t a0 = rk00, a1 = (J_) + rk01, b0 = rk10, b1 = (J_) + rk11;                                              \
    uint32_t c0 = rk00, c1 = (C1_) + rk01, d0 = rk10, d1 = (C1_) + rk11;                                            \
    uint64_t sA0, sB0, sA1, sB1;                                                                                    \
    if (one_gamete) {                                                                                               \
      asm volatile(BH_TFA                                                                                           \
                   : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)                                                         \
                   : "s"(ka.w[0]), "s"(ka.w[1]), "s"(ka.w[2]), "s"(ka.w[3]), "s"(ka.w[4]), "s"(ka.w[5]),            \
                     "s"(ka.w[6]), "s"(ka.w[7]));                                                                   \
      asm volatile(BH_TFA                                                                                           \
                   : "+v"(c0), "+v"(c1), "+v"(d0), "+v"(d1)                                                         \
                   : "s"(ka.w[0]), "s"(ka.w[1]), "s"(ka.w[2]), "s"(ka.w[3]), "s"(ka.w[4]), "s"(ka.w[5]),            \
                     "s"(ka.w[6]), "s"(ka.w[7]));                                                                   \
      sA0 = sA1 = __ballot((a0 ^ a1) < (TA_));                                                                      \
      sB0 = sB1 = __ballot((c0 ^ c1) < (TB_));                                                                      \
    } else {                                                                                                        \
      asm volatile(BH_TFA BH_TFB                                                                                    \
                   : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)                                                         \
                   : "s"(ka.w[0]), "s"(ka.w[1]), "s"(ka.w[2]), "s"(ka.w[3]), "s"(ka.w[4]), "s"(ka.w[5]),            \
                     "s"(ka.w[6]), "s"(ka.w[7]), "s"(kb.w[0]), "s"(kb.w[1]), "s"(kb.w[2]), "s"(kb.w[3]),            \
                     "s"(kb.w[4]), "s"(kb.w[5]), "s"(kb.w[6]), "s"(kb.w[7]));                                       \
      asm volatile(BH_TFA BH_TFB                                                                                    \
                   : "+v"(c0), "+v"(c1), "+v"(d0), "+v"(d1)                                                         \
                   : "s"(ka.w[0]), "s"(ka.w[1]), "s"(ka.w[2]), "s"(ka.w[3]), "s"(ka.w[4]), "s"(ka.w[5]),            \
                     "s"(ka.w[6]), "s"(ka.w[7]), "s"(kb.w[0]), "s"(kb.w[1]), "s"(kb.w[2]), "s"(kb.w[3]),            \
                     "s"(kb.w[4]), "s"(kb.w[5]), "s"(kb.w[6]), "s"(kb.w[7]));                                       \
      sA0 = __ballot((a0 ^ a1) < (TA_));                                                                            \
      sB0 = __ballot((c0 ^ c1) < (TB_));                                                                            \
      sA1 = __ballot((b0 ^ b1) < (TA_));                                                                            \
      sB1 = __ballot((d0 ^ d1) < (TB_));                                                                            \
    }                                                                                                               \
    if (p.alwA) {                                                                                                   \
      const uint64_t xa = p.alwA[(T_)], xb = p.alwB[(T_)];                                                          \
      sA0 |= xa;                                                                                                    \
      sA1 |= xa;                                                                                                    \
      sB0 |= xb;                                                                                                    \
      sB1 |= xb;                                                                                                    \
    }                                                                                                               \
    if (lane == 0) {                                                                                                \
      ulonglong2 *dstw = reinterpret_cast<ulonglong2 *>(S + 4 * (T_));                                              \
      dstw[0] = make_ulonglong2(sA0, sB0);                                                                          \
      dstw[1] = make_ulonglong2(sA1, sB1);                                                                          \
    }                                                                                                               \
  }

#define BH_TLOAD(r_, v_, s_) __builtin_amdgcn_raw_buffer_load_b32(r_, v_, s_, 0)
constexpr int kMaskNpf = 4;  // trips whose tables are requested ahead
  constexpr int NPF = kMaskNpf;
  if (partitionable)
  {
    int t = wv;
    uint32_t j = (uint32_t)wv * 64u + lane;
    uint32_t soff = (uint32_t)wv * 256u;
    constexpr uint32_t STEP = 256u * WAVES;
    // the tables of NPF trips ahead are in flight: next to row waves that keep the memory pipeline full a load
    // waits its turn for microseconds (profiles/r02_exp_c3_roles_pmc_priorities.txt)
    uint32_t tA[NPF], tB[NPF], cB[NPF];
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
      tA[u] = BH_TLOAD(rA, voff, soff + (uint32_t)u * STEP);
      tB[u] = BH_TLOAD(rB, voff, soff + (uint32_t)u * STEP);
      cB[u] = BH_TLOAD(rC, voff, soff + (uint32_t)u * STEP);
    }
    while (t < p.nwh) {
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        if (u == 0 || t + u * WAVES < p.nwh) BH_TRIP_P(t + u * WAVES, j + 64u * (uint32_t)(u * WAVES), tA[u], tB[u], cB[u])
        tA[u] = BH_TLOAD(rA, voff, soff + (uint32_t)(NPF + u) * STEP);
        tB[u] = BH_TLOAD(rB, voff, soff + (uint32_t)(NPF + u) * STEP);
        cB[u] = BH_TLOAD(rC, voff, soff + (uint32_t)(NPF + u) * STEP);
      }
      t += NPF * WAVES;
      j += 64u * (uint32_t)(NPF * WAVES);
      soff += (uint32_t)NPF * STEP;
    }
  }
  else
  {
    int t = wv;
    uint32_t j = (uint32_t)wv * 64u + lane;
    uint32_t soff = (uint32_t)wv * 256u;
    constexpr uint32_t STEP = 256u * WAVES;
    // the tables of NPF trips ahead are in flight: next to row waves that keep the memory pipeline full a load
    // waits its turn for microseconds (profiles/r02_exp_c3_roles_pmc_priorities.txt)
    uint32_t tA[NPF], tB[NPF], cB[NPF];
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
      tA[u] = BH_TLOAD(rA, voff, soff + (uint32_t)u * STEP);
      tB[u] = BH_TLOAD(rB, voff, soff + (uint32_t)u * STEP);
      cB[u] = BH_TLOAD(rC, voff, soff + (uint32_t)u * STEP);
    }
    while (t < p.nwh) {
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        if (u == 0 || t + u * WAVES < p.nwh) BH_TRIP(t + u * WAVES, j + 64u * (uint32_t)(u * WAVES), tA[u], tB[u], cB[u])
        tA[u] = BH_TLOAD(rA, voff, soff + (uint32_t)(NPF + u) * STEP);
        tB[u] = BH_TLOAD(rB, voff, soff + (uint32_t)(NPF + u) * STEP);
        cB[u] = BH_TLOAD(rC, voff, soff + (uint32_t)(NPF + u) * STEP);
      }
      t += NPF * WAVES;
      j += 64u * (uint32_t)(NPF * WAVES);
      soff += (uint32_t)NPF * STEP;
    }
  }
#undef BH_TRIP
#undef BH_TRIP_P
#undef BH_TLOAD
  if (tid == 0) {
    S[4 * p.nwh + 1] = 0;  // the word behind the last B words (funnel shifts read one ahead)
    S[4 * p.nwh + 3] = 0;
  }
  if constexpr (SYNC == SYNC_WAVE) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else if constexpr (SYNC == SYNC_BLOCK) {
    __syncthreads();
  } else {
    // one-way hand-off inside a workgroup (all of whose waves are resident): no wave ever waits
    // for wave 0, so the wait below always ends; it is bounded all the same
    if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (wv != 0) return;
    bool ok = false;
    for (int spin = 0; spin < (1 << 24); ++spin) {
      if (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= (unsigned)(WAVES * (iter + 1))) {
        ok = true;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (!ok && lane == 0 && p.err) __hip_atomic_fetch_or(p.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }

  // ---- phase B: crossover masks = inclusive prefix-XOR of the site strings
  if (wv == 0) {
    uint8_t *Z = p.Z + (int64_t)c * p.zpitch;
    uint32_t carry0 = 0, carry1 = 0;
    const uint64_t below = (1ull << lane) - 1ull;
    for (int w0 = 0; w0 < p.nw; w0 += 64) {
      const int w = w0 + lane;
      uint64_t x0 = 0, x1 = 0;
      if (w < p.nw) {
        x0 = prefix_xor64(site_word(S, 0, w, p.half));
        x1 = prefix_xor64(site_word(S, 1, w, p.half));
      }
      const uint64_t par0 = __ballot((x0 >> 63) != 0), par1 = __ballot((x1 >> 63) != 0);
      if ((__popcll(par0 & below) & 1u) ^ carry0) x0 = ~x0;
      if ((__popcll(par1 & below) & 1u) ^ carry1) x1 = ~x1;
      carry0 ^= __popcll(par0) & 1u;
      carry1 ^= __popcll(par1) & 1u;
      if (w < p.nw) {
        // bit-interleaved: marker m's gamete-0 bit at bit 2 (m % 8), its gamete-1 bit next to it, so
        // that markers 8v..8v+7 are one u16 at Z + 2v whose bit i belongs to byte i of genotype vector v
        uint4 z;
        z.x = spread16((uint32_t)x0 & 0xffffu) | (spread16((uint32_t)x1 & 0xffffu) << 1);
        z.y = spread16((uint32_t)x0 >> 16) | (spread16((uint32_t)x1 >> 16) << 1);
        z.z = spread16((uint32_t)(x0 >> 32) & 0xffffu) | (spread16((uint32_t)(x1 >> 32) & 0xffffu) << 1);
        z.w = spread16((uint32_t)(x0 >> 48)) | (spread16((uint32_t)(x1 >> 48)) << 1);
        *reinterpret_cast<uint4 *>(Z + 16 * (int64_t)w) = z;
      }
    }
    if constexpr (SYNC == SYNC_COUNTER) {
      if (lane == 0) __hip_atomic_store(done, (unsigned)(iter + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

// offspring dword k of one 16-byte vector: dword k holds markers 2k, 2k+1 as bytes
// [m0 a0, m0 a1, m1 a0, m1 a1]; output byte (marker, p) = parent p's byte (marker, mask_p).
// zz = the vector's 16 mask bits, bit i for output byte i.  The multiply drops bits 4k .. 4k+3 onto
// bit 0 of the selector bytes of a v_perm_b32 whose sources are {parent 1 dword, parent 0 dword}.
__device__ __forceinline__ uint32_t select_dword(uint32_t zz, int k, uint32_t aw, uint32_t bw) {
  const uint32_t nib = (zz >> (4 * k)) & 0xfu;
  return __builtin_amdgcn_perm(bw, aw, ((nib * 0x00204081u) & 0x01010101u) | 0x06020400u);
}

// the eight dosages (0, 1, 2) of one offspring vector as doubles: dosage of markers (2k, 2k+1) =
// bytes 0 and 2 of  w + (w >> 8)  (alleles are 0/1)
struct Dosage8 {
  double d[8];
};
__device__ __forceinline__ Dosage8 dosages(const uint4 o) {
  const uint32_t s0 = o.x + (o.x >> 8), s1 = o.y + (o.y >> 8);
  const uint32_t s2 = o.z + (o.z >> 8), s3 = o.w + (o.w >> 8);
  Dosage8 r;
  r.d[0] = (double)(s0 & 0xffu);
  r.d[1] = (double)((s0 >> 16) & 0xffu);
  r.d[2] = (double)(s1 & 0xffu);
  r.d[3] = (double)((s1 >> 16) & 0xffu);
  r.d[4] = (double)(s2 & 0xffu);
  r.d[5] = (double)((s2 >> 16) & 0xffu);
  r.d[6] = (double)(s3 & 0xffu);
  r.d[7] = (double)((s3 >> 16) & 0xffu);
  return r;
}

// A group of WAVES waves streams R consecutive rows at once: the marker-effect vector of a trip is
// loaded once and applied to all R rows (with one row per wave the effects were 2x the genotype
// bytes out of L2).  R > 1 only with WAVES == 1.
// LONG rows (HBM-resident sets) want more loads in flight: two vectors per row and trip.
template <int WAVES, int TT, int MODE, int R, bool LONG, int PRIO = 3>
__device__ __forceinline__ void row_role(const RowParams &p, const int block, const int wave, double *red,
                                         RewardLds *rw = nullptr) {
  static_assert(WAVES == 1 || WAVES == 4, "a group is one wave or the whole workgroup");
  static_assert(R == 1 || WAVES == 1, "several rows per group only for one-wave groups");
  constexpr int GROUPS = 4 / WAVES;
  constexpr int STRIDE = 64 * WAVES;
  constexpr int NT = TT > 0 ? TT : 1;
  const int lane = threadIdx.x & 63;
  const int grp = wave / WAVES, wv = wave % WAVES;
  const int tid = wv * 64 + lane;  // thread index inside the group
  // wave-uniform by construction; readfirstlane tells the compiler, so row addressing is scalar
  const int c0 = __builtin_amdgcn_readfirstlane((xcd_row(block, p.n_blocks) * GROUPS + grp) * R);
  if (c0 >= p.n_rows) return;  // only possible for WAVES == 1, whose waves never meet at a barrier

  const uint8_t *src0[R], *src1[R];
  const uint16_t *Z16[R];
  uint8_t *dst[R];
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
      src0[r] = p.src + ((int64_t)e * p.n_pool + i0) * p.pitch;
      src1[r] = p.src + ((int64_t)e * p.n_pool + i1) * p.pitch;
      Z16[r] = reinterpret_cast<const uint16_t *>(p.Z + (int64_t)(p.rng_shared ? i : c) * p.zpitch);
    } else if constexpr (MODE == MODE_GATHER) {
      src0[r] = p.src + (int64_t)norm_index(p.idx[c], p.n_pool) * p.pitch;
    } else {
      src0[r] = p.src + (int64_t)c * p.pitch;
    }
    dst[r] = (MODE == MODE_GEBV) ? nullptr : p.out + (int64_t)c * p.pitch;
  }

  // Row waves share their SIMD with VALU-bound mask waves, and VALU issue is arbitrated by priority
  // first.  Next to one-wave-per-cross mask groups (many keyed crosses: the launch is Threefry-bound)
  // the little arithmetic a row wave has between its loads must not queue behind Threefry, or the
  // memory pipes drain: PRIO 3.  Next to four-wave mask groups (few keyed crosses: the launch is
  // bound by the streaming) equal priority lets the mask waves use the row waves' stalls instead of
  // being held to the end: PRIO 0 (measured: profiles/r01_ab_row_pipelined_priority.txt).
  if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);

  // NC independent chains per sum.  Long rows (HBM-resident sets, few waves per SIMD doing the sums)
  // gain 8 % from four chains -- a single chain of dependent f64 FMAs waits out the FMA latency
  // eight times per vector; 10k-marker rows lose 2 % to the extra registers (tools/run_ab_chains.sh).
  constexpr int NC = !LONG ? 1 : (NT * R <= 4 ? 4 : (NT * R <= 8 ? 2 : 1));
  double acc[R][NT][NC];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < NC; ++k) acc[r][t][k] = 0.0;

  // VPT 16-byte vectors per lane and trip for each of the R rows; all genotype loads of a trip
  // are issued before the first use
  constexpr int VPT = (R == 1 || (R == 2 && LONG)) ? 2 : 1;
  for (int v = tid; v < p.nvec; v += VPT * STRIDE) {
    int vv[VPT];
    bool has[VPT];
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
      has[u] = v + u * STRIDE < p.nvec;
      vv[u] = has[u] ? v + u * STRIDE : v;
    }
    uint4 a[R][VPT], b[R][VPT];
    uint32_t z[R][VPT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < VPT; ++u) {
        a[r][u] = *reinterpret_cast<const uint4 *>(src0[r] + 16 * (int64_t)vv[u]);
        if constexpr (MODE == MODE_CROSS) {
          b[r][u] = *reinterpret_cast<const uint4 *>(src1[r] + 16 * (int64_t)vv[u]);
          z[r][u] = Z16[r][vv[u]];
        }
      }
    uint4 oo[R][VPT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < VPT; ++u) {
        uint4 o = a[r][u];
        if constexpr (MODE == MODE_CROSS) {
          o.x = select_dword(z[r][u], 0, a[r][u].x, b[r][u].x);
          o.y = select_dword(z[r][u], 1, a[r][u].y, b[r][u].y);
          o.z = select_dword(z[r][u], 2, a[r][u].z, b[r][u].z);
          o.w = select_dword(z[r][u], 3, a[r][u].w, b[r][u].w);
        }
        if constexpr (MODE != MODE_GEBV) {
          if (live[r] && has[u]) *reinterpret_cast<uint4 *>(dst[r] + 16 * (int64_t)vv[u]) = o;
        }
        oo[r][u] = o;
      }
    if constexpr (TT > 0) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        if (t < p.tcount) {
#pragma unroll
          for (int u = 0; u < VPT; ++u) {
            if (has[u]) {
              const double2 *ep =
                  reinterpret_cast<const double2 *>(p.eff + (int64_t)t * p.effPitch + 8 * (int64_t)vv[u]);
              const double2 e0 = ep[0], e1 = ep[1], e2 = ep[2], e3 = ep[3];
#pragma unroll
              for (int r = 0; r < R; ++r) {
                const Dosage8 d = dosages(oo[r][u]);
                const double e[8] = {e0.x, e0.y, e1.x, e1.y, e2.x, e2.y, e3.x, e3.y};
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[r][t][k % NC] = fma(d.d[k], e[k], acc[r][t][k % NC]);
              }
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
        double s = acc[r][t][0];
#pragma unroll
        for (int k = 1; k < NC; ++k) s += acc[r][t][k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        acc[r][t][0] = s;
      }
    if constexpr (WAVES > 1) {
      if (lane == 0)
        for (int t = 0; t < TT; ++t) red[wv * TT + t] = acc[0][t][0];
      __syncthreads();
      if (tid == 0)
        for (int t = 0; t < TT; ++t) {
          double s = 0.0;
          for (int w = 0; w < WAVES; ++w) s += red[w * TT + t];
          acc[0][t][0] = s;
        }
    }
    if (tid == 0) {
      int env0 = 0, expected = 0;
      if constexpr (MODE == MODE_CROSS)
        if (p.reward && WAVES == 1) reward_wg_layout(block, p.n_blocks, R, p.n_rows, p.n_out, env0, expected);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (live[r]) {
          float best = -INFINITY;
#pragma unroll
          for (int t = 0; t < TT; ++t)
            if (t < p.tcount) {
              const float g = (float)acc[r][t][0];
              p.gebv[(int64_t)(c0 + r) * p.T + p.t0 + t] = g;
              best = g != g ? best : fmaxf(best, g);
            }
          if constexpr (MODE == MODE_CROSS)
            if (p.reward) reward_row(WAVES == 1 ? rw : nullptr, p.reward, p.reward_pitch, env0, c0 + r, p.n_out, best);
        }
      if constexpr (MODE == MODE_CROSS)
        if (p.reward && WAVES == 1) reward_wave_done(rw, p.reward, p.reward_pitch, env0, expected);
    }
  }
}

// The row role of the fused launch for short rows next to four-wave mask groups (few keyed crosses:
// the reference's shared keys), software-pipelined.  gfx950 counts loads and
// stores in one in-order counter (vmcnt): a trip that loads after it stored can only wait for that
// load by also waiting for the store to be acknowledged.  Here the genotype vectors and mask words of
// trip t+1 are requested before trip t's stores, and the marker effects of trip t before those, so
// that no wait covers a store; only the genotype registers exist twice (18 VGPRs for two rows).  The
// body is branch-free so that the compiler's counter model stays exact: lanes past the end of the row
// redo the last vector, dead row slots redo row c0 -- the same bytes stored twice.
template <int TT, int R>
__device__ __forceinline__ void row_role_pipelined(const RowParams &p, const int block, const int wave,
                                                   RewardLds *rw = nullptr) {
  static_assert(TT == 1 || TT == 2, "effects are held in registers across the trip");
  const int lane = threadIdx.x & 63;
  const int c0 = __builtin_amdgcn_readfirstlane((xcd_row(block, p.n_blocks) * 4 + wave) * R);
  if (c0 >= p.n_rows) return;

  const uint8_t *src0[R], *src1[R];
  const uint16_t *Z16[R];
  uint8_t *dst[R];
  bool live[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    live[r] = c0 + r < p.n_rows;
    const int c = live[r] ? c0 + r : c0;
    const int e = c / p.n_out, i = c - e * p.n_out;
    const int i0 = norm_index(p.idx[2 * (int64_t)c], p.n_pool);
    const int i1 = norm_index(p.idx[2 * (int64_t)c + 1], p.n_pool);
    src0[r] = p.src + ((int64_t)e * p.n_pool + i0) * p.pitch;
    src1[r] = p.src + ((int64_t)e * p.n_pool + i1) * p.pitch;
    Z16[r] = reinterpret_cast<const uint16_t *>(p.Z + (int64_t)(p.rng_shared ? i : c) * p.zpitch);
    dst[r] = p.out + (int64_t)c * p.pitch;
  }
  // no s_setprio: next to four-wave mask groups, see row_role

  // GEBV in exact fixed point (see packed_stream_role in packed_kernels.h): three balanced base-65536
  // digits of round(e * 2^s_t), dosage pairs as the int16 halves of (w + (w >> 8)) & 0x00FF00FF, one
  // v_dot2_i32_i16 per pair and digit instead of conversions and f64 FMAs
  int acc[R][TT][3];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[r][t][j] = 0;

  const int last = p.nvec - 1;
  struct Vecs {
    uint4 a[R], b[R];
    uint32_t z[R];
  };
  auto load_vecs = [&](int base, Vecs &w) {
    int v = base + lane;
    v = v < last ? v : last;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      w.a[r] = *reinterpret_cast<const uint4 *>(src0[r] + 16 * (int64_t)v);
      w.b[r] = *reinterpret_cast<const uint4 *>(src1[r] + 16 * (int64_t)v);
      w.z[r] = Z16[r][v];
    }
  };
  auto trip = [&](int base, const Vecs &w, Vecs &next) {
    const int v = base + lane;
    const bool has = v < p.nvec;
    const int vv = v < last ? v : last;
    uint4 dg[TT][3];
#pragma unroll
    for (int q = 0; q < TT; ++q) {
      const uint4 *dp = p.dig + (int64_t)(p.t0 + q) * 3 * (p.nvec + 1) + vv;
      dg[q][0] = dp[0];
      dg[q][1] = dp[p.nvec + 1];
      dg[q][2] = dp[2 * (int64_t)(p.nvec + 1)];
    }
    load_vecs(base + 64, next);
    uint4 oo[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      uint4 o;
      o.x = select_dword(w.z[r], 0, w.a[r].x, w.b[r].x);
      o.y = select_dword(w.z[r], 1, w.a[r].y, w.b[r].y);
      o.z = select_dword(w.z[r], 2, w.a[r].z, w.b[r].z);
      o.w = select_dword(w.z[r], 3, w.a[r].w, w.b[r].w);
      *reinterpret_cast<uint4 *>(dst[r] + 16 * (int64_t)vv) = o;
      oo[r] = o;
    }
    if (has) {  // a lane that redid the last vector must not count it twice
      typedef short short2v __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t w[4] = {oo[r].x, oo[r].y, oo[r].z, oo[r].w};
        uint32_t pr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) pr[k] = (w[k] + (w[k] >> 8)) & 0x00FF00FFu;
#pragma unroll
        for (int q = 0; q < TT; ++q)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const uint32_t dd[4] = {dg[q][j].x, dg[q][j].y, dg[q][j].z, dg[q][j].w};
            int a = acc[r][q][j];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              a = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, pr[k]), __builtin_bit_cast(short2v, dd[k]), a, false);
            acc[r][q][j] = a;
          }
      }
    }
  };
  Vecs wa, wb;
  load_vecs(0, wa);
  // enter the loop with nothing pending, so that the loop head inherits the back edge's wait ("the
  // vectors, but not the stores behind them") rather than its merge with "the loads just issued"
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (int base = 0; base < p.nvec; base += 128) {  // two trips per turn, no branch between them
    trip(base, wa, wb);
    trip(base + 64, wb, wa);
  }

  int env0 = 0, expected = 0;
  if (p.reward) reward_wg_layout(block, p.n_blocks, R, p.n_rows, p.n_out, env0, expected);
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      long long s = (long long)acc[r][t][0] + (long long)acc[r][t][1] * 65536LL + (long long)acc[r][t][2] * 4294967296LL;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      if (lane == 0 && live[r]) {
        const float g = (float)((double)s * p.scale[p.t0 + t]);
        p.gebv[(int64_t)(c0 + r) * p.T + p.t0 + t] = g;
        if (p.reward) reward_row(rw, p.reward, p.reward_pitch, env0, c0 + r, p.n_out, g);
      }
    }
  if (lane == 0 && p.reward) reward_wave_done(rw, p.reward, p.reward_pitch, env0, expected);
}

template <int WAVES>
__global__ __launch_bounds__(256) void mask_kernel(const MaskParams p) {
  extern __shared__ __align__(16) uint8_t lds_raw[];
  mask_role<WAVES, WAVES == 1 ? SYNC_WAVE : SYNC_BLOCK>(p, (int)blockIdx.x, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lds_raw, nullptr);
}

template <int WAVES, int TT, int MODE, int R, bool LONG>
__global__ __launch_bounds__(256) void row_kernel(const RowParams p) {
  __shared__ double red[4 * (TT > 0 ? TT : 1)];
  __shared__ RewardLds rwl;
  prime_next_reward(p.reward_next, p.n_envs, p.reward_pitch);
  if (p.reward) {
    if (threadIdx.x == 0) reward_lds_init(&rwl);
    __syncthreads();
  }
  row_role<WAVES, TT, MODE, R, LONG>(p, (int)blockIdx.x, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), red,
                                     &rwl);
}

// One launch, two roles, for rows that one wave handles.  A workgroup is 8 waves: waves 0-3
// stream four rows of generation g (`rp`, with the masks the previous launch made), waves 4-7
// make masks of generation g+1 (`mp`): of four crosses (MW = 1, many crosses to do), or all four
// on one cross (MW = 4: few crosses, e.g. masks shared by all envs -- the short way keeps the
// mask work from outlasting the streaming it hides under).  A workgroup's waves are dealt to
// the CU's four SIMDs in turn, so each SIMD gets one memory-bound and one VALU-bound wave per
// workgroup whatever the dispatcher does with workgroups.  (Giving whole workgroups one role
// each was measured to put each role on half of the CUs only.)  Row and mask waves never wait
// for each other after the barrier at entry.
template <int TT, int MW, int R, bool LONG>
__global__ __launch_bounds__(512) void cross_kernel(const RowParams rp, const MaskParams mp) {
  extern __shared__ __align__(16) uint8_t lds_raw[];
  __shared__ unsigned counter, done;
  __shared__ RewardLds rwl;
  prime_next_reward(rp.reward_next, rp.n_envs, rp.reward_pitch);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: keep it scalar
  if (MW == 4 || rp.reward) {  // uniform: the whole workgroup meets here or nowhere
    if (threadIdx.x == 0) {
      counter = 0;
      done = 0;
      reward_lds_init(&rwl);
    }
    __syncthreads();
  }
  if (wave < 4) {
    if ((int)blockIdx.x < rp.n_blocks) {
      if constexpr ((TT == 1 || TT == 2) && !LONG && MW == 4)
        row_role_pipelined<TT, R>(rp, (int)blockIdx.x, wave, &rwl);
      else
        row_role<1, TT, MODE_CROSS, R, LONG, MW == 4 ? 0 : 3>(rp, (int)blockIdx.x, wave, nullptr, &rwl);
    }
  } else if constexpr (MW == 1) {
    if ((int)blockIdx.x < mp.n_blocks) mask_role<1, SYNC_WAVE>(mp, (int)blockIdx.x, wave - 4, lds_raw, nullptr);
  } else {
    // the grid may be shorter than the mask work (somewhat more keyed crosses than row workgroups):
    // every workgroup then takes several crosses in turn, so that no tail of mask-only workgroups
    // is left when the streaming is over (two_role_grid in breedhip.hip)
    int iter = 0;
    for (int mb = (int)blockIdx.x; mb < mp.n_blocks; mb += (int)gridDim.x, ++iter)
      mask_role<4, SYNC_COUNTER>(mp, mb, wave - 4, lds_raw, &counter, iter, &done);
  }
}

// reward[e] = max over the env's (individuals x traits) GEBV block (vec_env.py:97)
__global__ __launch_bounds__(256) void reward_max_kernel(const float *g, int64_t per_env, float *out, int64_t pitch) {
  __shared__ float part[4];
  const int e = blockIdx.x, tid = threadIdx.x;
  const float *ge = g + (int64_t)e * per_env;
  float m = -INFINITY;
  for (int64_t j = tid; j < per_env; j += 256) m = fmaxf(m, ge[j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((tid & 63) == 0) part[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) out[e * pitch] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}
}  // namespace bh
