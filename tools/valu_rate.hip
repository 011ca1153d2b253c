// Micro-benchmark: per-SIMD issue rate of the VALU ops the Threefry / select path is made of.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../breedgym_amd/csrc/threefry.h"

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned s) {
  unsigned a = threadIdx.x, b = a * 3 + s, c = a ^ s, d = a + 7 * s;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 1) { REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 2) { REP16(asm volatile("v_alignbit_b32 %0, %0, %0, 19\n v_alignbit_b32 %1, %1, %1, 19\n v_alignbit_b32 %2, %2, %2, 19\n v_alignbit_b32 %3, %3, %3, 19" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 3) { REP16(asm volatile("v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 4) { REP16(asm volatile("v_xad_u32 %0, %0, %4, %1\n v_xad_u32 %1, %1, %4, %2\n v_xad_u32 %2, %2, %4, %3\n v_xad_u32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 5) { REP16(asm volatile("v_perm_b32 %0, %0, %4, %1\n v_perm_b32 %1, %1, %4, %2\n v_perm_b32 %2, %2, %4, %3\n v_perm_b32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 6) { float fa = __uint_as_float(a), fb = __uint_as_float(b), fc = __uint_as_float(c), fd = __uint_as_float(d), fs = __uint_as_float(s);
      REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(fs));)
      a = __float_as_uint(fa); b = __float_as_uint(fb); c = __float_as_uint(fc); d = __float_as_uint(fd); }
    if (OP == 7) { REP16(asm volatile("v_lshrrev_b32 %0, 9, %0\n v_lshrrev_b32 %1, 9, %1\n v_lshrrev_b32 %2, 9, %2\n v_lshrrev_b32 %3, 9, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 8) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 9) { REP16(asm volatile("v_and_or_b32 %0, %0, %4, %1\n v_and_or_b32 %1, %1, %4, %2\n v_and_or_b32 %2, %2, %4, %3\n v_and_or_b32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 10) { REP16(asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 11) { // dependent chain of one register: add -> alignbit -> xor
      REP16(asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n v_add_u32 %0, %0, %1" : "+v"(a), "+v"(b) : "v"(s));) }
    if (OP == 12) { double da = a, db = b, dc = c, dd = d, ds = s;
      REP16(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(ds));)
      a = (unsigned)da; b = (unsigned)db; c = (unsigned)dc; d = (unsigned)dd; }
    if (OP == 13) { REP16(asm volatile("v_bfe_u32 %0, %0, 3, 2\n v_bfe_u32 %1, %1, 3, 2\n v_bfe_u32 %2, %2, 3, 2\n v_bfe_u32 %3, %3, 3, 2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}

// the phase-A inner loop without memory: two Threefry calls + 4 compares per iteration
template <int VAR>
__global__ __launch_bounds__(256) void tfk(unsigned long long *out, int iters, unsigned k0, unsigned k1, unsigned thr) {
  unsigned j = threadIdx.x & 63;
  unsigned long long acc = 0;
  for (int i = 0; i < iters; ++i, j += 64) {
    bh::u32x2 a = bh::threefry2x32(k0, k1, j, j + 5000u);
    bh::u32x2 b = bh::threefry2x32(k1, k0, j, j + 5000u);
    if (VAR == 0) {
      acc ^= __ballot((a.x >> 9) < thr) ^ __ballot((a.y >> 9) < thr) ^ __ballot((b.x >> 9) < thr) ^ __ballot((b.y >> 9) < thr);
    } else {
      acc ^= (unsigned long long)(a.x ^ a.y ^ b.x ^ b.y);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int VAR>
void run_tf(const char *name, int waves_per_simd) {
  const int iters = 790;
  const int blocks = 256 * waves_per_simd;
  unsigned long long *out;
  hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  tfk<VAR><<<blocks, 256>>>(out, 10, 1, 2, 12000);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  tfk<VAR><<<blocks, 256>>>(out, iters, 1, 2, 12000);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double cyc = ms * 1e-3 * 2.4e9;
  printf("%-16s waves/SIMD=%d  %.3f ms  -> %.0f cycles per (2 Threefry + compares) iteration per wave-slot\n", name, waves_per_simd, ms,
         cyc / ((double)waves_per_simd * iters));
  hipFree(out);
}

template <int OP>
void run(const char *name, int waves_per_simd) {
  const int iters = 2000;
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per 256-thread block) = waves_per_simd per SIMD
  unsigned *out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(out, 10, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(out, iters, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // instructions per SIMD = waves_per_simd * iters * 64
  double inst = (double)waves_per_simd * iters * 64;
  double cyc = ms * 1e-3 * 2.4e9;
  printf("%-16s waves/SIMD=%d  %.3f ms  -> %.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc / inst);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4, 8}) { run_tf<0>("tf2+ballot", w); run_tf<1>("tf2+xor", w); }
  for (int w : {4}) {
    run<0>("v_add_u32", w); run<1>("v_xor_b32", w); run<2>("v_alignbit_b32", w); run<3>("v_add3_u32", w);
    run<4>("v_xad_u32", w); run<5>("v_perm_b32", w); run<6>("v_fma_f32", w); run<7>("v_lshrrev_b32", w);
    run<8>("v_mul_u32_u24", w); run<9>("v_and_or_b32", w); run<10>("v_pk_add_u16", w); run<11>("tf_round_dep", w);
    run<12>("v_fma_f64", w); run<13>("v_bfe_u32", w);
  }
  return 0;
}
