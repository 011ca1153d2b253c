// Where does an s_nop (or a second independent chain) recover the VALU issue slots that a
// dependent add -> alignbit -> xor chain loses?  gfx950, 8 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(x) x x x x x x x x
template <int P>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned long long *clk) {
  unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 77, d = a + 99;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (P == 0) asm volatile(R8("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n") : "+v"(a), "+v"(b));
    if (P == 1) asm volatile(R8("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n s_nop 0\n") : "+v"(a), "+v"(b));
    if (P == 2) asm volatile(R8("v_add_u32 %0, %0, %1\n s_nop 0\n v_alignbit_b32 %1, %1, %1, 19\n s_nop 0\n v_xor_b32 %1, %1, %0\n s_nop 0\n") : "+v"(a), "+v"(b));
    if (P == 3) asm volatile(R8("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n s_nop 0\n v_xor_b32 %1, %1, %0\n") : "+v"(a), "+v"(b));
    if (P == 4) asm volatile(R8("v_add_u32 %0, %0, %1\n s_nop 0\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n") : "+v"(a), "+v"(b));
    if (P == 5) asm volatile(R8("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_alignbit_b32 %1, %1, %1, 19\n v_alignbit_b32 %3, %3, %3, 19\n v_xor_b32 %1, %1, %0\n v_xor_b32 %3, %3, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if (P == 6) asm volatile(R8("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_alignbit_b32 %1, %1, %1, 19\n v_alignbit_b32 %3, %3, %3, 19\n v_xor_b32 %1, %1, %0\n v_xor_b32 %3, %3, %2\n s_nop 0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if (P == 7) asm volatile(R8("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n s_nop 1\n") : "+v"(a), "+v"(b));
    if (P == 8) asm volatile(R8("v_alignbit_b32 %1, %1, %1, 19\n v_add_u32 %0, %0, %1\n v_xor_b32 %1, %1, %0\n") : "+v"(a), "+v"(b));
    if (P == 9) asm volatile(R8("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n s_mov_b32 s20, s21\n") : "+v"(a), "+v"(b) :: "s20");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
template <int P>
void run(const char *name, int w, int nvalu) {
  const int iters = 4000, blocks = 256 * w;
  unsigned *out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  unsigned long long *clk, h[2]; (void)hipMalloc(&clk, 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<P><<<blocks, 256>>>(out, 10, clk); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<P><<<blocks, 256>>>(out, iters, clk); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / (double)h[1] * 0.1;
  printf("%-46s w=%d %.3f ms clock %.2f GHz -> %.2f cycles per VALU instr, %.1f cycles per round\n", name, w, ms, ghz,
         ms * 1e-3 * ghz * 1e9 / ((double)w * iters * nvalu), ms * 1e-3 * ghz * 1e9 / ((double)w * iters * nvalu) * 3);
  (void)hipFree(out);
}
int main() {
  for (int w : {8, 4, 2}) {
    run<0>("add,alignbit,xor", w, 24); run<1>("add,alignbit,xor,s_nop", w, 24); run<2>("s_nop after every instr", w, 24);
    run<3>("add,alignbit,s_nop,xor", w, 24); run<4>("add,s_nop,alignbit,xor", w, 24);
    run<5>("2 chains interleaved", w, 48); run<6>("2 chains interleaved + s_nop", w, 48); run<7>("round + s_nop 1", w, 24);
    run<8>("alignbit,add,xor", w, 24); run<9>("round + s_mov", w, 24);
  }
}
