// Full Threefry-2x32-20 x 2 chains (the phase-A inner loop body): which hand-placed
// instruction stream is fastest on a gfx950 SIMD?  Compared with the compiler's own stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../breedgym_amd/csrc/threefry.h"

// operands: %0,%1 = chain A x0,x1; %2,%3 = chain B x0,x1; %4..%11 = A key words; %12..%19 = B key words
//   key words per chain: [0]=k1 [1]=k2+1 [2]=k2 [3]=k0+2 [4]=k0 [5]=k1+3 [6]=k2+4 [7]=k0+5
#define RA(r) "v_add_u32 %0, %0, %1\n s_nop 0\n v_alignbit_b32 %1, %1, %1, " #r "\n v_xor_b32 %1, %1, %0\n"
#define RB(r) "v_add_u32 %2, %2, %3\n s_nop 0\n v_alignbit_b32 %3, %3, %3, " #r "\n v_xor_b32 %3, %3, %2\n"
#define IA(ka, kb) "v_add_u32 %0, " ka ", %0\n v_add_u32 %1, " kb ", %1\n"
#define IB(ka, kb) "v_add_u32 %2, " ka ", %2\n v_add_u32 %3, " kb ", %3\n"
#define TFA RA(19) RA(17) RA(6) RA(26) IA("%4", "%5") RA(15) RA(3) RA(16) RA(8) IA("%6", "%7") \
            RA(19) RA(17) RA(6) RA(26) IA("%8", "%9") RA(15) RA(3) RA(16) RA(8) IA("%4", "%10") \
            RA(19) RA(17) RA(6) RA(26) IA("%6", "%11")
#define TFB RB(19) RB(17) RB(6) RB(26) IB("%12", "%13") RB(15) RB(3) RB(16) RB(8) IB("%14", "%15") \
            RB(19) RB(17) RB(6) RB(26) IB("%16", "%17") RB(15) RB(3) RB(16) RB(8) IB("%12", "%18") \
            RB(19) RB(17) RB(6) RB(26) IB("%14", "%19")
// alternating rounds of A and B
#define RAB(r) RA(r) RB(r)
#define IAB(a1, a2, b1, b2) IA(a1, a2) IB(b1, b2)
#define TFAB RAB(19) RAB(17) RAB(6) RAB(26) IAB("%4", "%5", "%12", "%13") RAB(15) RAB(3) RAB(16) RAB(8) IAB("%6", "%7", "%14", "%15") \
             RAB(19) RAB(17) RAB(6) RAB(26) IAB("%8", "%9", "%16", "%17") RAB(15) RAB(3) RAB(16) RAB(8) IAB("%4", "%10", "%12", "%18") \
             RAB(19) RAB(17) RAB(6) RAB(26) IAB("%6", "%11", "%14", "%19")
// no nops at all (reference point)
#define NA(r) "v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, " #r "\n v_xor_b32 %1, %1, %0\n"
#define NB(r) "v_add_u32 %2, %2, %3\n v_alignbit_b32 %3, %3, %3, " #r "\n v_xor_b32 %3, %3, %2\n"
#define TFNA NA(19) NA(17) NA(6) NA(26) IA("%4", "%5") NA(15) NA(3) NA(16) NA(8) IA("%6", "%7") \
             NA(19) NA(17) NA(6) NA(26) IA("%8", "%9") NA(15) NA(3) NA(16) NA(8) IA("%4", "%10") \
             NA(19) NA(17) NA(6) NA(26) IA("%6", "%11")
#define TFNB NB(19) NB(17) NB(6) NB(26) IB("%12", "%13") NB(15) NB(3) NB(16) NB(8) IB("%14", "%15") \
             NB(19) NB(17) NB(6) NB(26) IB("%16", "%17") NB(15) NB(3) NB(16) NB(8) IB("%12", "%18") \
             NB(19) NB(17) NB(6) NB(26) IB("%14", "%19")
// nop after the injection pair as well
#define IA2(ka, kb) "v_add_u32 %0, " ka ", %0\n s_nop 0\n v_add_u32 %1, " kb ", %1\n"
#define IB2(ka, kb) "v_add_u32 %2, " ka ", %2\n s_nop 0\n v_add_u32 %3, " kb ", %3\n"
#define TFA2 RA(19) RA(17) RA(6) RA(26) IA2("%4", "%5") RA(15) RA(3) RA(16) RA(8) IA2("%6", "%7") \
             RA(19) RA(17) RA(6) RA(26) IA2("%8", "%9") RA(15) RA(3) RA(16) RA(8) IA2("%4", "%10") \
             RA(19) RA(17) RA(6) RA(26) IA2("%6", "%11")
#define TFB2 RB(19) RB(17) RB(6) RB(26) IB2("%12", "%13") RB(15) RB(3) RB(16) RB(8) IB2("%14", "%15") \
             RB(19) RB(17) RB(6) RB(26) IB2("%16", "%17") RB(15) RB(3) RB(16) RB(8) IB2("%12", "%18") \
             RB(19) RB(17) RB(6) RB(26) IB2("%14", "%19")

template <int VAR>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned k0, unsigned k1, unsigned long long *clk) {
  unsigned j = threadIdx.x & 63, acc = 0;
  const unsigned k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  const unsigned A0 = k1, A1 = k2 + 1, A2 = k2, A3 = k0 + 2, A4 = k0, A5 = k1 + 3, A6 = k2 + 4, A7 = k0 + 5;
  const unsigned B0 = k0, B1 = k2 + 1, B2 = k2, B3 = k1 + 2, B4 = k1, B5 = k0 + 3, B6 = k2 + 4, B7 = k1 + 5;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i, j += 64) {
    unsigned a0 = j + k0, a1 = j + 5000u + k1, b0 = j + k1, b1 = j + 5000u + k0;
#define OPS : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) : "s"(A0), "s"(A1), "s"(A2), "s"(A3), "s"(A4), "s"(A5), "s"(A6), "s"(A7), \
              "s"(B0), "s"(B1), "s"(B2), "s"(B3), "s"(B4), "s"(B5), "s"(B6), "s"(B7)
    if (VAR == 0) { bh::u32x2 a = bh::threefry2x32(k0, k1, j, j + 5000u), b = bh::threefry2x32(k1, k0, j, j + 5000u); a0 = a.x; a1 = a.y; b0 = b.x; b1 = b.y; }
    if (VAR == 1) asm volatile(TFNA TFNB OPS);
    if (VAR == 2) asm volatile(TFA TFB OPS);
    if (VAR == 3) asm volatile(TFAB OPS);
    if (VAR == 4) asm volatile(TFA2 TFB2 OPS);
    acc ^= a0 ^ a1 ^ b0 ^ b1;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int VAR>
void run(const char *name, int w) {
  const int iters = 790, blocks = 256 * w;
  unsigned *out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  unsigned long long *clk, h[2]; (void)hipMalloc(&clk, 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<VAR><<<blocks, 256>>>(out, 10, 1, 2, clk); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<VAR><<<blocks, 256>>>(out, iters, 1, 2, clk); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / (double)h[1] * 0.1;
  unsigned first; (void)hipMemcpy(&first, out, 4, hipMemcpyDeviceToHost);
  printf("%-34s w=%d %.3f ms clock %.2f GHz -> %.0f cycles, %.1f ns per 2 Threefry per SIMD  [chk %08x]\n", name, w, ms, ghz,
         ms * 1e-3 * ghz * 1e9 / ((double)w * iters), ms * 1e6 / ((double)w * iters), first);
  (void)hipFree(out);
}

int main() {
  for (int w : {4, 8}) {
    run<0>("compiler", w); run<1>("asm A;B no nops", w); run<2>("asm A;B nop/round", w); run<3>("asm A/B alternating nop/round", w);
    run<4>("asm A;B nop/round + nop in inj", w);
  }
}
