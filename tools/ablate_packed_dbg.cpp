// Phase timeline of packed_cross_kernel<1,4,2> at BASELINE configs[1] (a variant built by tools/ab_build.sh with
// s_memtime stamps in it; the sed expressions are in profiles/r04_timeline_c2.txt).  Per workgroup: thread 0 (a row wave)
// stamps entry / behind the entry barrier / parent pointers ready / first words in / stream done / GEBV + reward noted;
// thread 256 (a mask wave) stamps its role's entry / key schedule made / sites drawn / masks written.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "breedhip.h"
extern "C" int bh_debug_dump(unsigned long long *host, int n);
int main() {
  const int64_t M = 10000, E = 16, n = 256, rows = E * n;
  std::vector<float> r(M, 0.0015f), eff(M, 1.0f);
  for (int64_t m = 0; m < M; m += M / 10) r[m] = 0.5f;
  bh_ctx *ctx;
  if (bh_create(0, M, 1, r.data(), eff.data(), &ctx)) { printf("create: %s\n", bh_last_error()); return 1; }
  const int64_t pitch = bh_row_pitch(ctx), pp = bh_packed_pitch(ctx);
  uint8_t *bytes, *ob; uint16_t *pool, *out; int32_t *idx; float *gebv, *rw[2];
  bh_alloc(ctx, rows * pitch, (void **)&bytes); bh_alloc(ctx, rows * pitch, (void **)&ob);
  bh_alloc(ctx, rows * pp, (void **)&pool); bh_alloc(ctx, rows * pp, (void **)&out);
  bh_alloc(ctx, rows * 2 * 4, (void **)&idx); bh_alloc(ctx, rows * 4, (void **)&gebv);
  bh_alloc(ctx, E * 64 * 4, (void **)&rw[0]); bh_alloc(ctx, E * 64 * 4, (void **)&rw[1]);
  std::vector<uint8_t> h(rows * pitch);
  { uint64_t s = 88172645463325252ull; uint64_t *w = reinterpret_cast<uint64_t *>(h.data());
    for (size_t i = 0; i < h.size() / 8; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w[i] = s & 0x0101010101010101ull; } }
  std::vector<int32_t> hi(rows * 2); for (auto &v : hi) v = rand() % (int)n;
  bh_memcpy_h2d(ctx, bytes, h.data(), h.size()); bh_memcpy_h2d(ctx, idx, hi.data(), hi.size() * 4);
  bh_pack(ctx, bytes, rows, pool); bh_sync(ctx);
  uint32_t key[2] = {0, 7};
  for (int i = 0; i < 20; ++i)
    bh_cross_packed(ctx, i & 1 ? out : pool, E, n, idx, n, key, 0, i & 1 ? pool : out, ob, gebv, rw[(i + 1) & 1], rw[i & 1], 64);
  bh_sync(ctx);
  const int W = 1024, S = 16;
  std::vector<unsigned long long> d(W * S);
  bh_debug_dump(d.data(), W * S);
  const char *names[10] = {"entry", "behind the entry barrier", "row: parent pointers ready", "row: first words in", "row: stream done",
                           "row: GEBV + reward noted", "mask: role entered", "mask: key schedule made", "mask: sites drawn", "mask: masks written"};
  printf("packed_cross_kernel<1,4,2>, 16 x 256 x 10k, the 20th launch of a loop: shader clocks (~2.1 per ns) since the workgroup's entry\n");
  for (int k = 1; k < 10; ++k) {
    unsigned long long lo = ~0ull, hi = 0; double sum = 0; int n = 0;
    for (int w = 0; w < W; ++w) if (d[w * S] && d[w * S + k]) { const unsigned long long v = d[w * S + k] - d[w * S]; lo = std::min(lo, v); hi = std::max(hi, v); sum += (double)v; ++n; }
    if (n) printf("  %-28s %4d workgroups  min %6llu  mean %8.1f  max %6llu\n", names[k], n, lo, sum / n, hi);
  }
  for (int half = 0; half < 2; ++half) {  // workgroups with mask work (the first 256) and without
    double s4 = 0, s5 = 0; unsigned long long hi5 = 0; int n = 0;
    for (int w = half * 256; w < half * 256 + 256; ++w) if (d[w * S]) { s4 += (double)(d[w * S + 4] - d[w * S + 3]); s5 += (double)(d[w * S + 5] - d[w * S]); hi5 = std::max(hi5, d[w * S + 5] - d[w * S]); ++n; }
    printf("  workgroups %3d..%3d (%s mask work): stream %8.1f clocks, life %8.1f mean, %6llu max\n", half * 256, half * 256 + 255, half ? "no" : "with", s4 / n, s5 / n, hi5);
  }
  {  // life by position in the XCD's dispatch order: workgroup b is the (b / 8)-th of XCD b % 8
    for (int q = 0; q < 64; q += 8) {
      double s5 = 0; int n = 0;
      for (int w = 0; w < 512; ++w) if ((w >> 3) >= q && (w >> 3) < q + 8 && d[w * S]) { s5 += (double)(d[w * S + 5] - d[w * S]); ++n; }
      printf("  dispatch positions %2d..%2d of each XCD: life %8.1f clocks (%d workgroups)\n", q, q + 7, s5 / n, n);
    }
  }
  return 0;
}
