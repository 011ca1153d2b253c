// The founder draw of VecBreedGym.reset (breedgym/vector/vec_env.py:22-27,120-128) on the device, small germplasm:
// one workgroup per env, JAX's sort-based shuffle inside LDS.  A role (device function) so that it can ride on the
// horizon step's two-role launch (packed_kernels.h) as well as run as a kernel of its own (select_kernels.h).
// Germplasm beyond one workgroup's LDS takes the radix sort in HBM (sort_kernels.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "threefry.h"

namespace bh {

// Ascending bitonic sort of P (a power of two) 64-bit keys in LDS by the whole workgroup.  Pair j of a stage touches
// elements of ONE aligned block of 2 * stride keys, and the 64 pairs a wave takes in one turn are consecutive: while
// stride <= 64 a wave stays inside its own aligned block of 128 keys, which nobody else touches in that stage -- and a
// wave's LDS operations are served in order.  So a workgroup barrier is needed only around the stages that reach
// further (15 of the 78 stages of a 4096-key sort); the others are separated by a wave-level fence.
__device__ __forceinline__ void bitonic_sort_lds(uint64_t *keys, int P) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int j = tid; j < (P >> 1); j += nt) {
        const int lo = ((j & ~(stride - 1)) << 1) | (j & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const uint64_t a = keys[lo], b = keys[hi];
        if ((a > b) == up) {
          keys[lo] = b;
          keys[hi] = a;
        }
      }
      const int next = stride > 1 ? stride >> 1 : size;  // the stride of the stage that follows
      if (stride > 64 || next > 64) {
        __syncthreads();
      } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
  __syncthreads();
}

// ---- founder draw of VecBreedGym.reset on the device ---------------------------------------
// env e: key_e = split(env_key, E+1)[1+e]; permutation(key_e, N)[:n] by jax's `_shuffle`:
// `rounds` times { key, sub = split(key); stable-sort the current order by random_bits(sub, N) }.
// One workgroup per env; the sort is a bitonic sort of (random word << 32 | position) in LDS,
// which is the stable sort because positions are unique.  N <= 8192.
struct FoundersParams {
  int32_t *out;  // (E, n)
  uint32_t key0, key1;
  int32_t E, N, n, P, rounds;
  int32_t layout;  // LAYOUT_ORIGINAL / LAYOUT_PARTITIONABLE
  int32_t E_total, env_offset;  // env e here is env env_offset + e of E_total (bh_set_env_window)
};

// the draw of env e by one workgroup of any size; fk: 16 P bytes of LDS (P keys, then 2 x P int32 orders)
__device__ __forceinline__ void founders_role(const FoundersParams &p, const int e, uint64_t *fk) {
  int32_t *x = reinterpret_cast<int32_t *>(fk + p.P), *y = x + p.P;
  const int tid = threadIdx.x, nt = blockDim.x;
  u32x2 k = split_at(p.key0, p.key1, (uint32_t)(p.E_total + 1), (uint32_t)(1 + p.env_offset + e), p.layout);
  for (int i = tid; i < p.N; i += nt) x[i] = i;
  const int L = p.N + (p.N & 1), h = L >> 1;
  for (int rd = 0; rd < p.rounds; ++rd) {
    const u32x2 nk = split_at(k.x, k.y, 2u, 0u, p.layout), sub = split_at(k.x, k.y, 2u, 1u, p.layout);
    k = nk;
    if (p.layout == LAYOUT_PARTITIONABLE) {
      for (int i = tid; i < p.N; i += nt)
        fk[i] = ((uint64_t)bits_word_partitionable(sub.x, sub.y, (uint32_t)i) << 32) | (uint32_t)i;
    } else {
      for (int i = tid; i < h; i += nt) {  // random_bits(sub, N): odd N is padded with a 0 counter
        const bool second = i + h < p.N;
        const u32x2 o = threefry2x32(sub.x, sub.y, (uint32_t)i, second ? (uint32_t)(i + h) : 0u);
        fk[i] = ((uint64_t)o.x << 32) | (uint32_t)i;
        if (second) fk[i + h] = ((uint64_t)o.y << 32) | (uint32_t)(i + h);
      }
    }
    for (int i = p.N + tid; i < p.P; i += nt) fk[i] = ~0ull;
    __syncthreads();
    bitonic_sort_lds(fk, p.P);
    for (int i = tid; i < p.N; i += nt) y[i] = x[(uint32_t)fk[i]];
    __syncthreads();
    int32_t *t = x;
    x = y;
    y = t;
  }
  for (int i = tid; i < p.n; i += nt) p.out[(int64_t)e * p.n + i] = x[i];
}

}  // namespace bh
