// Mutation: chromax's `Simulator(mutation=p)`, which the reference forwards through **kwargs
// (breedgym/vector/vec_env.py:39,45; breedgym/breedgym.py:74-80).  [EXTERNAL/UNVERIFIED, as the rest of the
// meiosis: chromax is not in the reference tree]  Inside `_meiosis` the gamete key is split in two,
//   rec_key, mut_key = split(gamete_key, 2)
// the crossover mask comes from rec_key (row_kernels.h / mask_stream.h), and every allele of the finished gamete is
// flipped where  bernoulli(mut_key, p, (M,)) = uniform(mut_key, (M,)) < p.
//
// p = 0 is the default and the hot path: nothing here runs then.  With p > 0 the cross launch streams the
// un-mutated offspring without GEBV, `mutation_sites_kernel` draws the flips of every keyed gamete into bit
// arrays (one ballot per 64 draws), `mutation_apply_kernel` XORs them into the offspring rows -- bit planes and /
// or byte layout, touching only the words that have a flip -- and the GEBV / reward passes run behind it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "threefry.h"

namespace bh {

struct MutationParams {
  uint64_t *F;         // (n_streams, fpitch) flip words: [0, wl) markers [0, half), [wl, 2 wl) markers [half, M)
  int64_t fpitch;      // 64-bit words per stream: 2 * wl + 1 (the last one stays zero: 16-bit windows read past)
  int32_t M, half, wl; // half = ceil(M / 2) in the original layout (draws come in pairs i, i + half), M otherwise
  int32_t n_streams;   // keyed gametes: 2 B, or B for double haploids
  uint32_t key0, key1; // the call's simulator key
  uint32_t nkeys;      // keys split off it (the whole batch's: sharding windows)
  uint32_t key_offset; // cross b of this call is cross key_offset + b of that batch
  int32_t dh, layout;
  uint32_t thr;        // uniform_f32(bits) < p  <=>  (bits >> 9) < thr
};

// grid (x: chunks of 64 draws in turns, y: keyed gamete), 256 threads
__global__ __launch_bounds__(256) void mutation_sites_kernel(MutationParams p) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const uint32_t s = blockIdx.y;
  const uint32_t kidx = p.dh ? 1u + p.key_offset + s : 1u + 2u * (p.key_offset + (s >> 1)) + (s & 1u);
  // mut_key = split(gamete_key, 2)[1]
  u32x2 mk;
  if (p.layout == LAYOUT_PARTITIONABLE) {
    const u32x2 gk = threefry2x32(p.key0, p.key1, 0u, kidx);
    mk = threefry2x32(gk.x, gk.y, 0u, 1u);
  } else {
    const uint32_t g0 = iota_word_even(p.key0, p.key1, 2u * p.nkeys, 2u * kidx);
    const uint32_t g1 = iota_word_even(p.key0, p.key1, 2u * p.nkeys, 2u * kidx + 1u);
    mk = {threefry2x32(g0, g1, 0u, 2u).y, threefry2x32(g0, g1, 1u, 3u).y};  // words 2, 3 of threefry_iota(gk, 4)
  }
  uint64_t *F = p.F + (int64_t)s * p.fpitch;
  // the draws: Threefry and little else (as many calls as the crossover masks of the same crosses take) -- two chunks of
  // 64 draws per wave and turn through the hand-placed two-chain block of threefry.h, the key schedule in scalar
  // registers (round 5: 2048 x 2 x 100k draws took 920 us with one chain of the compiler's stream per lane)
  const uint32_t k0 = __builtin_amdgcn_readfirstlane(mk.x), k1 = __builtin_amdgcn_readfirstlane(mk.y);
  const KeySched ks = key_sched(k0, k1);
  for (int chunk = (blockIdx.x * 4 + wid) * 2; chunk < p.wl; chunk += gridDim.x * 8) {
    const bool two = chunk + 1 < p.wl;  // (uniform; a wave whose second chunk does not exist redoes its first)
    const uint32_t ia = (uint32_t)chunk * 64u + lane, ib = two ? ia + 64u : ia;
    uint64_t wa0, wa1, wb0 = 0, wb1 = 0;
    if (p.layout == LAYOUT_PARTITIONABLE) {
      uint32_t a0 = k0, a1 = ia + k1, b0 = k0, b1 = ib + k1;  // random_bits(mk, M)[i] = x ^ y of threefry(mk, (0, i))
      threefry2x32_x2(ks, a0, a1, b0, b1);
      wa0 = __ballot(ia < (uint32_t)p.M && ((a0 ^ a1) >> 9) < p.thr);
      wa1 = __ballot(ib < (uint32_t)p.M && ((b0 ^ b1) >> 9) < p.thr);  // (no second half in this layout: wb = 0)
    } else {
      const uint32_t ja = ia + (uint32_t)p.half, jb = ib + (uint32_t)p.half;
      uint32_t a0 = ia + k0, a1 = (ja < (uint32_t)p.M ? ja : 0u) + k1;  // the pad counter of an odd M is 0
      uint32_t b0 = ib + k0, b1 = (jb < (uint32_t)p.M ? jb : 0u) + k1;
      threefry2x32_x2(ks, a0, a1, b0, b1);
      wa0 = __ballot(ia < (uint32_t)p.half && (a0 >> 9) < p.thr);
      wb0 = __ballot(ja < (uint32_t)p.M && (a1 >> 9) < p.thr);
      wa1 = __ballot(ib < (uint32_t)p.half && (b0 >> 9) < p.thr);
      wb1 = __ballot(jb < (uint32_t)p.M && (b1 >> 9) < p.thr);
    }
    if (lane == 0) {
      F[chunk] = wa0;
      F[p.wl + chunk] = wb0;
      if (two) {
        F[chunk + 1] = wa1;
        F[p.wl + chunk + 1] = wb1;
      }
    }
  }
}

struct MutationApplyParams {
  const uint64_t *F;
  int64_t fpitch;
  int32_t M, half, wl;
  int32_t n_rows, n_out;  // offspring rows; rows per env
  int32_t shared;         // 1: row c uses the streams of cross c % n_out, 0: of cross c
  int32_t dh;             // 1: one stream per cross, flipping both alleles alike
  uint16_t *outp;         // (n_rows, ppitch) bit-plane rows or null
  int64_t ppitch;         // in 16-bit words
  uint8_t *outb;          // (n_rows, bpitch) byte-layout rows or null
  int64_t bpitch;
};

// One thread per 64-bit flip word of a stream: almost all of them are zero (p = 1e-4: ten flips per 100k-marker
// gamete), and a thread whose word is zero is done after one load.  A set bit is a marker of one gamete of one cross:
// it flips that allele in every offspring row that uses the cross's streams (with shared keys: one row per env).  The
// two gametes of a cross (and the two halves of a row's markers) meet in the same 16-bit bit-plane words from
// different threads: the bit planes are flipped with 32-bit atomic XORs; in the byte layout every (marker, allele) is
// a byte of its own.  (Round 5: this replaces a thread per (ROW, eight markers) -- 102 M threads at 8192 x 100k, each
// reading its window of flip bits: 274 us; by flip words the same work is 6.4 M threads.)
__global__ __launch_bounds__(256) void mutation_apply_kernel(MutationApplyParams p, int32_t n_streams) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int32_t per = 2 * p.wl;
  if (t >= (int64_t)n_streams * per) return;
  const int32_t s = (int32_t)(t / per), q = (int32_t)(t - (int64_t)s * per);
  uint64_t w = p.F[(int64_t)s * p.fpitch + q];
  if (w == 0) return;
  const int32_t b = p.dh ? s : s >> 1, g = p.dh ? 0 : (s & 1);
  const int32_t m0 = q < p.wl ? 64 * q : p.half + 64 * (q - p.wl);
  while (w) {
    const int32_t m = m0 + __builtin_ctzll(w);
    w &= w - 1;
    if (m >= p.M) break;  // (the draws stop at M: nothing is set there)
    const uint32_t bits = (p.dh ? 3u : 1u << g) << (2 * (m & 7));
    for (int32_t row = b; row < p.n_rows; row += p.n_out) {  // the rows of the cross: one, or one per env (shared keys)
      if (p.outp) {
        uint16_t *hw = p.outp + (int64_t)row * p.ppitch + (m >> 3);
        uint32_t *wd = reinterpret_cast<uint32_t *>(reinterpret_cast<uintptr_t>(hw) & ~(uintptr_t)3);
        atomicXor(wd, bits << ((reinterpret_cast<uintptr_t>(hw) & 2) ? 16 : 0));
      }
      if (p.outb) {
        uint8_t *o = p.outb + (int64_t)row * p.bpitch + 2 * (int64_t)m;
        if (p.dh) {
          o[0] ^= 1;
          o[1] ^= 1;
        } else {
          o[g] ^= 1;
        }
      }
      if (!p.shared) break;
    }
  }
}

}  // namespace bh
