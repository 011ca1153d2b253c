// Timing driver for the radix sort in HBM through the C ABI: bh_selection_actions at configs[3]'s shape (65,536 scores,
// k = 6,553: 21.5 M pair ranks shuffled in three rounds) and a founder draw of n of N.
// usage: driver [n] [k] [n_crosses]     (tools/ab_build.sh --driver tools/ablate_sort.cpp builds variants)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "breedhip.h"

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 65536, k = argc > 2 ? atoll(argv[2]) : 6553, nc = argc > 3 ? atoll(argv[3]) : 8192;
  std::vector<float> r(64, 0.01f), eff(64, 1.0f);
  bh_ctx *ctx;
  if (bh_create(0, 64, 1, r.data(), eff.data(), &ctx)) { printf("create: %s\n", bh_last_error()); return 1; }
  std::vector<float> h(n);
  uint64_t s = 88172645463325252ull;
  for (auto &x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (float)(s % 1000003) * 1e-3f; }
  float *scores; int32_t *act;
  bh_alloc(ctx, n * 4, (void **)&scores); bh_alloc(ctx, nc * 8, (void **)&act);
  bh_memcpy_h2d(ctx, scores, h.data(), n * 4); bh_sync(ctx);
  uint32_t key[2] = {0, 9};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) if (bh_selection_actions(ctx, scores, 1, n, k, nc, key, act)) { printf("%s\n", bh_last_error()); return 1; }
  hipDeviceSynchronize();
  const int reps = 8;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) bh_selection_actions(ctx, scores, 1, n, k, nc, key, act);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("selection_actions n=%lld k=%lld (%lld pairs) n_crosses=%lld : %.1f us per call\n", (long long)n, (long long)k,
         (long long)(k * (k - 1) / 2), (long long)nc, ms / reps * 1e3);
  return 0;
}
