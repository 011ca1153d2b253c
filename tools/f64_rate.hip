// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the ops a GEBV accumulation can be built from.
// Build: hipcc -O3 --offload-arch=gfx950 -o f64_rate f64_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned s) {
  unsigned a = threadIdx.x, b = a * 3 + s, c = a ^ s, d = a + 7 * s;
  double da = a, db = b, dc = c, dd = d, ds = s;
  float fa = a, fb = b, fc = c, fd = d;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP16(asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %5\n v_cvt_f64_u32 %2, %6\n v_cvt_f64_u32 %3, %7" : "=v"(da), "=v"(db), "=v"(dc), "=v"(dd) : "v"(a), "v"(b), "v"(c), "v"(d));) }
    if (OP == 1) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(ds));) }
    if (OP == 2) { REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(ds));) }
    if (OP == 3) { REP16(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "=v"(da), "=v"(db), "=v"(dc), "=v"(dd) : "v"(fa), "v"(fb), "v"(fc), "v"(fd));) }
    if (OP == 4) { REP16(asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte2 %1, %5\n v_cvt_f32_ubyte0 %2, %6\n v_cvt_f32_ubyte2 %3, %7" : "=v"(fa), "=v"(fb), "=v"(fc), "=v"(fd) : "v"(a), "v"(b), "v"(c), "v"(d));) }
    if (OP == 5) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s) : "vcc");) }
    if (OP == 6) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 7) { REP16(asm volatile("v_perm_b32 %0, %0, %4, %1\n v_perm_b32 %1, %1, %4, %2\n v_perm_b32 %2, %2, %4, %3\n v_perm_b32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 8) { REP16(asm volatile("v_dot4_i32_i8 %0, %0, %4, %1\n v_dot4_i32_i8 %1, %1, %4, %2\n v_dot4_i32_i8 %2, %2, %4, %3\n v_dot4_i32_i8 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 9) { REP16(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 10) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 20, %4\n v_lshl_or_b32 %1, %1, 20, %4\n v_lshl_or_b32 %2, %2, 20, %4\n v_lshl_or_b32 %3, %3, 20, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 11) { REP16(asm volatile("v_dot2_i32_i16 %0, %0, %4, %1\n v_dot2_i32_i16 %1, %1, %4, %2\n v_dot2_i32_i16 %2, %2, %4, %3\n v_dot2_i32_i16 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));) }
    if (OP == 12) { REP16(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(s), "v"(a));) }
    if (OP == 13) { REP16(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(ds));) }
    if (OP == 14) { unsigned long long la = a, lb = b, lc = c, ld = d;
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(la), "+v"(lb), "+v"(lc), "+v"(ld) : "v"(s), "v"(a) : "vcc");)
      a ^= (unsigned)la ^ (unsigned)lb ^ (unsigned)lc ^ (unsigned)ld; }
    if (OP == 15) { REP16(asm volatile("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %6\n v_cvt_f64_i32 %3, %7" : "=v"(da), "=v"(db), "=v"(dc), "=v"(dd) : "v"(a), "v"(b), "v"(c), "v"(d));) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (unsigned)da ^ (unsigned)db ^ (unsigned)dc ^ (unsigned)dd ^ (unsigned)fa ^ (unsigned)fb ^ (unsigned)fc ^ (unsigned)fd;
}

template <int OP>
void run(const char *name, int wps) {
  const int iters = 1000, blocks = 256 * wps;
  unsigned *out;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(out, 10, 3);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(out, iters, 3);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)iters * 64 * wps;  // wave-instructions per SIMD
  printf("%-18s waves/SIMD=%d : %.2f ns per wave-instr per SIMD  (= %.1f cycles at 2.1 GHz)\n", name, wps, ms * 1e6 / instr, ms * 1e6 / instr * 2.1);
  (void)hipFree(out);
}

int main() {
  for (int w : {1, 4}) {
    run<0>("v_cvt_f64_u32", w); run<15>("v_cvt_f64_i32", w); run<1>("v_fma_f64", w); run<2>("v_add_f64", w); run<13>("v_mul_f64", w);
    run<3>("v_cvt_f64_f32", w); run<4>("v_cvt_f32_ubyteN", w); run<12>("v_fmac_f32", w); run<5>("v_cndmask_b32", w);
    run<6>("v_mad_u32_u24", w); run<7>("v_perm_b32", w); run<8>("v_dot4_i32_i8", w); run<11>("v_dot2_i32_i16", w);
    run<9>("v_and_b32", w); run<10>("v_lshl_or_b32", w); run<14>("v_mad_u64_u32", w);
  }
  return 0;
}
