// Census: how many 256-thread workgroups of a trivial ALU kernel are resident per CU at once?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
struct Rec { unsigned long long t0, t1; unsigned hwid, xcc; };
__global__ __launch_bounds__(256) void k(Rec *rec, unsigned *out, int iters) {
  unsigned a = threadIdx.x, b = a * 3 + 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n" : "+v"(a), "+v"(b));
  }
  out[blockIdx.x * 256 + threadIdx.x] = a ^ b;
  if (threadIdx.x == 0) {
    Rec r; r.t0 = t0; r.t1 = __builtin_amdgcn_s_memrealtime();
    r.hwid = __builtin_amdgcn_s_getreg(63492); r.xcc = __builtin_amdgcn_s_getreg(63508);
    rec[blockIdx.x] = r;
  }
}
int main(int argc, char **argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 2048, iters = 100000;
  Rec *d; unsigned *out; (void)hipMalloc(&d, blocks * sizeof(Rec)); (void)hipMalloc(&out, (size_t)blocks * 1024);
  k<<<blocks, 256>>>(d, out, 10); (void)hipDeviceSynchronize();
  k<<<blocks, 256>>>(d, out, iters); (void)hipDeviceSynchronize();
  std::vector<Rec> r(blocks); (void)hipMemcpy(r.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull, tmax = 0; for (auto &x : r) { tmin = std::min(tmin, x.t0); tmax = std::max(tmax, x.t1); }
  std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;  // per CU events
  for (auto &x : r) { unsigned cu = ((x.xcc & 15) << 16) | (x.hwid & 0xFF00) | ((x.hwid >> 13) & 7) << 20; ev[cu].push_back({x.t0, 1}); ev[cu].push_back({x.t1, -1}); }
  int maxc = 0; double avgmax = 0; std::map<int,int> hist;
  for (auto &kv : ev) { auto v = kv.second; std::sort(v.begin(), v.end()); int c = 0, m = 0; for (auto &e : v) { c += e.second; m = std::max(m, c); } maxc = std::max(maxc, m); avgmax += m; hist[m]++; }
  printf("blocks=%d distinct CU ids=%zu  kernel span %.1f us, first block %.1f us; max resident blocks/CU=%d avg-of-max=%.2f\n", blocks, ev.size(),
         (tmax - tmin) / 100.0, (r[0].t1 - r[0].t0) / 100.0, maxc, avgmax / ev.size());
  for (auto &h : hist) printf("  CUs with max %d resident blocks: %d\n", h.first, h.second);
  printf("hwid sample %08x %08x %08x xcc %x %x\n", r[0].hwid, r[1].hwid, r[8].hwid, r[0].xcc, r[1].xcc);
}
