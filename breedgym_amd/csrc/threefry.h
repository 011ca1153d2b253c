// Threefry-2x32-20 and the JAX key layouts built on it, shared by host and device code.
//
// This is the PRNG the reference's arithmetic runs on (chromax -> jax.random, default
// "threefry2x32" implementation, original non-partitionable layouts).  The layouts:
//   threefry_iota(key, n)[i]: counters iota(n) padded to even length L with a 0, cut into
//     halves x0 = c[0:h], x1 = c[h:L] (h = L/2), hashed pairwise; word i comes from
//       i <  h : TF(key, (i, i+h < n ? i+h : 0)).o0
//       i >= h : TF(key, (i-h, i)).o1
//   split(key, num)      = threefry_iota(key, 2*num) reshaped (num, 2)
//   random_bits(key, n)  = threefry_iota(key, n)
// Those are JAX's ORIGINAL layouts (jax_threefry_partitionable = False: the default up to jax 0.4.x, under
// which the reference's seeded expectations were recorded).  Since jax 0.5 the default is the
// PARTITIONABLE form (jax/_src/prng.py, _threefry_split_foldlike / _threefry_random_bits_partitionable):
// counters are the 64-bit positions (hi, lo) = (0, i), one Threefry call per element,
//   split(key, num)[k]      = TF(key, (0, k))            both output words are the new key
//   random_bits(key, n)[i]  = o0 ^ o1 of TF(key, (0, i))  (32-bit draws)
// Both are selectable (bh_set_threefry_layout); everything built on split / random_bits follows.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BH_HD __host__ __device__ __forceinline__
#else
#define BH_HD inline
#endif

// On the device, hide a value's provenance from the optimiser (zero instructions).  Without it
// LLVM sees that the counter is affine in the loop index, keeps that affine part out of the
// x0 += x1 chain and pays two adds per round instead of one.
#if defined(__HIP_DEVICE_COMPILE__)
#define BH_OPAQUE(v) asm volatile("" : "+v"(v))
#else
#define BH_OPAQUE(v) (void)(v)
#endif

namespace bh {

struct u32x2 {
  uint32_t x, y;
};

BH_HD uint32_t rotl32(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }

#define BH_TF_ROUND(r) \
  x0 += x1;            \
  x1 = rotl32(x1, r);  \
  x1 ^= x0;

BH_HD u32x2 threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  uint32_t x0 = c0 + k0, x1 = c1 + k1;
  BH_OPAQUE(x0);
  BH_OPAQUE(x1);
  BH_TF_ROUND(13) BH_TF_ROUND(15) BH_TF_ROUND(26) BH_TF_ROUND(6)
  x0 += k1; x1 += k2 + 1u;
  BH_TF_ROUND(17) BH_TF_ROUND(29) BH_TF_ROUND(16) BH_TF_ROUND(24)
  x0 += k2; x1 += k0 + 2u;
  BH_TF_ROUND(13) BH_TF_ROUND(15) BH_TF_ROUND(26) BH_TF_ROUND(6)
  x0 += k0; x1 += k1 + 3u;
  BH_TF_ROUND(17) BH_TF_ROUND(29) BH_TF_ROUND(16) BH_TF_ROUND(24)
  x0 += k1; x1 += k2 + 4u;
  BH_TF_ROUND(13) BH_TF_ROUND(15) BH_TF_ROUND(26) BH_TF_ROUND(6)
  x0 += k2; x1 += k0 + 5u;
  return {x0, x1};
}

// word j of threefry_iota(key, n) for EVEN n (the only case split() needs)
BH_HD uint32_t iota_word_even(uint32_t k0, uint32_t k1, uint32_t n, uint32_t j) {
  const uint32_t h = n >> 1;
  const bool first = j < h;
  const uint32_t c0 = first ? j : j - h;
  const u32x2 o = threefry2x32(k0, k1, c0, c0 + h);
  return first ? o.x : o.y;
}

enum { LAYOUT_ORIGINAL = 0, LAYOUT_PARTITIONABLE = 1 };

// split(key, num)[k] without materialising the other keys
BH_HD u32x2 split_at(uint32_t k0, uint32_t k1, uint32_t num, uint32_t k, int layout = LAYOUT_ORIGINAL) {
  if (layout == LAYOUT_PARTITIONABLE) return threefry2x32(k0, k1, 0u, k);
  return {iota_word_even(k0, k1, 2u * num, 2u * k), iota_word_even(k0, k1, 2u * num, 2u * k + 1u)};
}

// random_bits(key, n)[i] in the partitionable layout (the original one hashes positions in pairs: see the
// loops that use it)
BH_HD uint32_t bits_word_partitionable(uint32_t k0, uint32_t k1, uint32_t i) {
  const u32x2 o = threefry2x32(k0, k1, 0u, i);
  return o.x ^ o.y;
}

// integer threshold t with  uniform_f32(bits) < r  <=>  (bits >> 9) < t   (t in [0, 2^23])
inline uint32_t threshold23(float r) {
  if (!(r > 0.0f)) return 0u;
  if (r >= 1.0f) return 1u << 23;
  double x = (double)r * 8388608.0;  // exact: r is f32, 2^23 a power of two
  uint32_t t = (uint32_t)x;
  return ((double)t < x) ? t + 1u : t;  // ceil
}

}  // namespace bh
