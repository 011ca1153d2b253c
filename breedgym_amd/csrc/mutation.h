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
  for (int chunk = blockIdx.x * 4 + wid; chunk < p.wl; chunk += gridDim.x * 4) {
    const uint32_t i = (uint32_t)chunk * 64u + lane;
    if (p.layout == LAYOUT_PARTITIONABLE) {
      const bool flip = i < (uint32_t)p.M && (bits_word_partitionable(mk.x, mk.y, i) >> 9) < p.thr;
      const uint64_t w = __ballot(flip);
      if (lane == 0) {
        F[chunk] = w;
        F[p.wl + chunk] = 0;  // no second half in this layout
      }
    } else {
      const uint32_t j = i + (uint32_t)p.half;
      const u32x2 o = threefry2x32(mk.x, mk.y, i, j < (uint32_t)p.M ? j : 0u);  // the pad counter of an odd M is 0
      const uint64_t wa = __ballot(i < (uint32_t)p.half && (o.x >> 9) < p.thr);
      const uint64_t wb = __ballot(j < (uint32_t)p.M && (o.y >> 9) < p.thr);
      if (lane == 0) {
        F[chunk] = wa;
        F[p.wl + chunk] = wb;
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

// the flips of markers [m0, m0 + 8) of one stream; m0 a multiple of 8
__device__ __forceinline__ uint32_t mutation_flips8(const uint64_t *F, int32_t m0, int32_t half, int32_t wl) {
  const uint8_t *lo = reinterpret_cast<const uint8_t *>(F), *hi = reinterpret_cast<const uint8_t *>(F + wl);
  auto window = [](const uint8_t *a, int32_t bit) -> uint32_t {
    const uint32_t v = (uint32_t)a[bit >> 3] | ((uint32_t)a[(bit >> 3) + 1] << 8);
    return (v >> (bit & 7)) & 0xFFu;
  };
  uint32_t f = 0;
  if (m0 < half) f = window(lo, m0);  // bits at and past `half` are zero there
  if (m0 + 8 > half) f |= m0 >= half ? window(hi, m0 - half) : (window(hi, 0) << (half - m0)) & 0xFFu;
  return f;
}

__device__ __forceinline__ uint32_t spread8(uint32_t f) {  // bit k -> bit 2k
  f = (f | (f << 4)) & 0x0F0Fu;
  f = (f | (f << 2)) & 0x3333u;
  return (f | (f << 1)) & 0x5555u;
}

// one thread per (row, 8 markers)
__global__ __launch_bounds__(256) void mutation_apply_kernel(MutationApplyParams p) {
  const int32_t nw8 = (p.M + 7) >> 3;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)p.n_rows * nw8) return;
  const int32_t row = (int32_t)(t / nw8), w = (int32_t)(t - (int64_t)row * nw8);
  const int32_t b = p.shared ? row % p.n_out : row;
  const uint64_t *F0 = p.F + (int64_t)(p.dh ? b : 2 * b) * p.fpitch;
  const uint32_t f0 = mutation_flips8(F0, 8 * w, p.half, p.wl);
  const uint32_t f1 = p.dh ? f0 : mutation_flips8(F0 + p.fpitch, 8 * w, p.half, p.wl);
  if ((f0 | f1) == 0) return;
  if (p.outp) p.outp[(int64_t)row * p.ppitch + w] ^= (uint16_t)(spread8(f0) | (spread8(f1) << 1));
  if (p.outb) {
    uint8_t *o = p.outb + (int64_t)row * p.bpitch + 16 * (int64_t)w;
    for (int k = 0; k < 8; ++k) {
      if ((f0 >> k) & 1u) o[2 * k] ^= 1;
      if ((f1 >> k) & 1u) o[2 * k + 1] ^= 1;
    }
  }
}

}  // namespace bh
