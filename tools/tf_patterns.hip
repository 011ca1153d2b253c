// What does one Threefry round cost?  Dependent instruction patterns, 4 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int P>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned s, unsigned long long *clk) {
  unsigned a = threadIdx.x, b = a * 3 + s;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (P == 1) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b));) }
    if (P == 2) { asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n"
                               "v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 17\n v_xor_b32 %1, %1, %0\n"
                               "v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 6\n v_xor_b32 %1, %1, %0\n"
                               "v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 26\n v_xor_b32 %1, %1, %0\n"
                               "v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 15\n v_xor_b32 %1, %1, %0\n"
                               "v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 3\n v_xor_b32 %1, %1, %0\n"
                               "v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 16\n v_xor_b32 %1, %1, %0\n"
                               "v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 8\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b)); }
    if (P == 3) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %1, %1, %0\n v_add_u32 %0, %0, %1\n" : "+v"(a), "+v"(b));) }
    if (P == 4) { REP8(asm volatile("v_alignbit_b32 %1, %1, %1, 19\n v_alignbit_b32 %0, %0, %0, 19\n v_alignbit_b32 %1, %1, %1, 19\n" : "+v"(a), "+v"(b));) }
    if (P == 5) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b) : "v"(s));) }
    if (P == 6) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_lshlrev_b32 %2, 13, %1\n v_lshrrev_b32 %1, 19, %1\n v_or_b32 %1, %1, %2\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b), "+v"(s));) }
    if (P == 10) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b));) }
    if (P == 11) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 17\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b));) }
    if (P == 12) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 6\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b));) }
    if (P == 13) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 26\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b));) }
    if (P == 14) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 15\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b));) }
    if (P == 15) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 3\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b));) }
    if (P == 16) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 16\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b));) }
    if (P == 17) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 8\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b));) }
    if (P == 20) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, %2\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b) : "v"(s));) }
    if (P == 21) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, %2\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b) : "s"(s));) }
    if (P == 30) { asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b)); asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 17\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b)); asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 6\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b)); asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 26\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b)); asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 15\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b)); asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 3\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b)); asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 16\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b)); asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 8\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b)); }
    if (P == 31) { asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n" "v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n" "v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n" "v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n" "v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n" "v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n" "v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n" "v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b)); }
    if (P == 7) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %1, %1, %0\n v_alignbit_b32 %1, %1, %1, 19\n" : "+v"(a), "+v"(b));) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
template <int P>
void run(const char *name, int w, int ninst) {
  const int iters = 4000, blocks = 256 * w;
  unsigned *out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  unsigned long long *clk, hclk[2]; (void)hipMalloc(&clk, 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<P><<<blocks, 256>>>(out, 10, 1, clk); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<P><<<blocks, 256>>>(out, iters, 1, clk); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)hclk[0] / (double)hclk[1] * 0.1;
  printf("%-44s w=%d %.3f ms  clock %.2f GHz -> %.2f cycles/instr (block 0: %.2f)\n", name, w, ms, ghz,
         ms * 1e-3 * ghz * 1e9 / ((double)w * iters * ninst), (double)hclk[0] / ((double)iters * ninst));
  (void)hipFree(out);
}
int main() {
  for (int w : {8}) {
    run<10>("round, rot const 19", w, 24);
    run<11>("round, rot const 17", w, 24);
    run<12>("round, rot const 6", w, 24);
    run<13>("round, rot const 26", w, 24);
    run<14>("round, rot const 15", w, 24);
    run<15>("round, rot const 3", w, 24);
    run<16>("round, rot const 16", w, 24);
    run<17>("round, rot const 8", w, 24);
    run<30>("8 stmts, 8 consts", w, 24); run<31>("1 stmt, same const", w, 24); run<2>("1 stmt, 8 consts", w, 24); run<1>("8 stmts same const", w, 24);
    run<20>("round, rot in VGPR", w, 24); run<21>("round, rot in SGPR", w, 24);
  }
  for (int w : {8}) { if (w) continue;
    run<1>("add,alignbit,xor (same rot)", w, 24); run<2>("add,alignbit,xor (8 rots)", w, 24); run<3>("add,xor,add", w, 24);
    run<4>("alignbit x3", w, 24); run<6>("add,shl,shr,or,xor", w, 40); run<7>("add,xor,alignbit", w, 24);
  }
}
