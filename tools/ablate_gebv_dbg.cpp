// Phase timeline of gebv_bits_kernel (a variant built by tools/ab_build.sh with stamps in it: see the sed expressions in
// profiles/r04_ab_gebv.txt): per workgroup five s_memtime stamps -- start, first rows landed, main loop done, first
// additions returned, end -- printed as min / mean / max offsets from the launch's earliest start.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "breedhip.h"
extern "C" int bh_debug_dump(unsigned long long *host, int n);
int main(int argc, char **argv) {
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 2048, M = argc > 2 ? atoll(argv[2]) : 100000, T = argc > 3 ? atoll(argv[3]) : 8;
  std::vector<float> r(M, 0.0015f), eff(M * T);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (auto &e : eff) e = (float)((int64_t)(rnd() % 2001) - 1000) * 0.01f;
  bh_ctx *ctx;
  if (bh_create(0, M, T, r.data(), eff.data(), &ctx)) { printf("create: %s\n", bh_last_error()); return 1; }
  const int64_t pp = bh_packed_pitch(ctx);
  uint16_t *pool; float *gebv;
  bh_alloc(ctx, rows * pp, (void **)&pool); bh_alloc(ctx, rows * T * 4, (void **)&gebv);
  std::vector<uint64_t> h(rows * pp / 8); for (auto &w : h) w = rnd() & rnd();
  bh_memcpy_h2d(ctx, pool, h.data(), h.size() * 8); bh_sync(ctx);
  for (int i = 0; i < 5; ++i) bh_gather_packed(ctx, pool, rows, nullptr, rows, nullptr, nullptr, gebv);
  bh_sync(ctx);
  const int W = 1024, S = 8;
  std::vector<unsigned long long> d(W * S);
  bh_debug_dump(d.data(), W * S);
  int n = 0;
  for (int w = 0; w < W; ++w) n += d[w * S] != 0;
  const char *names[5] = {"start", "rows landed", "main loop done", "additions returned", "end"};
  printf("rows=%lld M=%lld T=%lld: %d workgroups; s_memtime ticks since the workgroup's own start (the XCDs' counters are not aligned)\n", (long long)rows, (long long)M, (long long)T, n);
  for (int k = 1; k < 5; ++k) {
    unsigned long long lo = ~0ull, hi = 0; double sum = 0;
    for (int w = 0; w < W; ++w) if (d[w * S]) { const unsigned long long v = d[w * S + k] - d[w * S]; lo = std::min(lo, v); hi = std::max(hi, v); sum += (double)v; }
    printf("  %-20s min %7llu  mean %9.1f  max %7llu\n", names[k], lo, sum / n, hi);
  }
  // start skew inside one XCD (workgroups b, b + 8, ... share a counter)
  for (int x = 0; x < 2; ++x) {
    unsigned long long lo = ~0ull, hi = 0, ehi = 0;
    for (int w = x; w < W; w += 8) if (d[w * S]) { lo = std::min(lo, d[w * S]); hi = std::max(hi, d[w * S]); ehi = std::max(ehi, d[w * S + 4]); }
    printf("  XCD class %d: starts spread over %llu ticks, last end %llu ticks after the first start\n", x, hi - lo, ehi - lo);
  }
  return 0;
}
