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

#if defined(__HIPCC__)
// ---- Threefry-2x32-20 for two key pairs at once, hand-placed ----------------------------------
// Measured on gfx950 (tools/tf_variants.hip, tools/tf_nops.hip): a dependent
// v_add_u32 -> v_alignbit_b32 -> v_xor_b32 stream issues at ~13 cycles per round, and at ~9 when
// one scalar no-op follows the add, for 2..8 waves per SIMD alike; hipcc's own stream for the
// same arithmetic takes ~25 % longer than this one.  Operands: %0,%1 = x0,x1 of chain A,
// %2,%3 = chain B; per chain eight scalar key words
//   k1, k2+1, k2, k0+2, k0, k1+3, k2+4, k0+5      (k2 = k0 ^ k1 ^ 0x1BD11BDA)
// The caller has already added (k0, k1) to the counters.
#define BH_RA(r) "v_add_u32 %0, %0, %1\n s_nop 0\n v_alignbit_b32 %1, %1, %1, " #r "\n v_xor_b32 %1, %1, %0\n"
#define BH_RB(r) "v_add_u32 %2, %2, %3\n s_nop 0\n v_alignbit_b32 %3, %3, %3, " #r "\n v_xor_b32 %3, %3, %2\n"
#define BH_IA(ka, kb) "v_add_u32 %0, " ka ", %0\n s_nop 0\n v_add_u32 %1, " kb ", %1\n"
#define BH_IB(ka, kb) "v_add_u32 %2, " ka ", %2\n s_nop 0\n v_add_u32 %3, " kb ", %3\n"
// rotate-left by R is v_alignbit by 32 - R: {13,15,26,6} -> {19,17,6,26}; {17,29,16,24} -> {15,3,16,8}
#define BH_TFA                                                                                          \
  BH_RA(19) BH_RA(17) BH_RA(6) BH_RA(26) BH_IA("%4", "%5") BH_RA(15) BH_RA(3) BH_RA(16) BH_RA(8) BH_IA("%6", "%7")  \
  BH_RA(19) BH_RA(17) BH_RA(6) BH_RA(26) BH_IA("%8", "%9") BH_RA(15) BH_RA(3) BH_RA(16) BH_RA(8) BH_IA("%4", "%10") \
  BH_RA(19) BH_RA(17) BH_RA(6) BH_RA(26) BH_IA("%6", "%11")
#define BH_TFB                                                                                                  \
  BH_RB(19) BH_RB(17) BH_RB(6) BH_RB(26) BH_IB("%12", "%13") BH_RB(15) BH_RB(3) BH_RB(16) BH_RB(8) BH_IB("%14", "%15") \
  BH_RB(19) BH_RB(17) BH_RB(6) BH_RB(26) BH_IB("%16", "%17") BH_RB(15) BH_RB(3) BH_RB(16) BH_RB(8) BH_IB("%12", "%18") \
  BH_RB(19) BH_RB(17) BH_RB(6) BH_RB(26) BH_IB("%14", "%19")

struct KeySched {  // the eight scalar words one chain needs
  uint32_t w[8];
};
__device__ __forceinline__ KeySched key_sched(uint32_t k0, uint32_t k1) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  return {{k1, k2 + 1u, k2, k0 + 2u, k0, k1 + 3u, k2 + 4u, k0 + 5u}};
}

// Two Threefry-2x32-20 calls under ONE key whose schedule is wave-uniform (scalar registers): (a0, a1) and (b0, b1) hold
// counter + (k0, k1) going in and the two output words coming out.  The draws of the shuffles (shuffle_sparse.h) run
// this instead of the compiler's stream for threefry2x32(): ~25 % fewer cycles per call (tools/tf_variants.hip).
__device__ __forceinline__ void threefry2x32_x2(const KeySched &k, uint32_t &a0, uint32_t &a1, uint32_t &b0, uint32_t &b1) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass of hipcc sees the declaration only)
  asm volatile(BH_TFA BH_TFB
               : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)
               : "s"(k.w[0]), "s"(k.w[1]), "s"(k.w[2]), "s"(k.w[3]), "s"(k.w[4]), "s"(k.w[5]), "s"(k.w[6]), "s"(k.w[7]),
                 "s"(k.w[0]), "s"(k.w[1]), "s"(k.w[2]), "s"(k.w[3]), "s"(k.w[4]), "s"(k.w[5]), "s"(k.w[6]), "s"(k.w[7]));
#endif
}
#endif  // __HIPCC__

// integer threshold t with  uniform_f32(bits) < r  <=>  (bits >> 9) < t   (t in [0, 2^23])
inline uint32_t threshold23(float r) {
  if (!(r > 0.0f)) return 0u;
  if (r >= 1.0f) return 1u << 23;
  double x = (double)r * 8388608.0;  // exact: r is f32, 2^23 a power of two
  uint32_t t = (uint32_t)x;
  return ((double)t < x) ? t + 1u : t;  // ceil
}

}  // namespace bh
