// Ablation driver for the fused cross kernel at BASELINE configs[1] size, through the C ABI.
// Build three times (see tools/run_ablation.sh): full, -DBH_ABLATE_NO_A (no Threefry loop),
// -DBH_ABLATE_NO_C (no streaming loop).  Timing only -- ablated builds give wrong offspring.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "breedhip.h"

int main(int argc, char **argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int64_t M = argc > 2 ? atoll(argv[2]) : 10000, E = argc > 3 ? atoll(argv[3]) : 16, n = argc > 4 ? atoll(argv[4]) : 256;
  const int reps = 50;
  std::vector<float> r(M, 0.0015f), eff(M, 1.0f);
  for (int64_t m = 0; m < M; m += M / 10) r[m] = 0.5f;
  bh_ctx *ctx;
  if (bh_create(0, M, 1, r.data(), eff.data(), &ctx)) { printf("create: %s\n", bh_last_error()); return 1; }
  const int64_t pitch = bh_row_pitch(ctx), rows = E * n;
  uint8_t *pool, *out; int32_t *idx; float *gebv;
  bh_alloc(ctx, rows * pitch, (void **)&pool); bh_alloc(ctx, rows * pitch, (void **)&out);
  bh_alloc(ctx, rows * 2 * 4, (void **)&idx); bh_alloc(ctx, rows * 4, (void **)&gebv);
  std::vector<uint8_t> h(rows * pitch); for (auto &b : h) b = rand() & 1;
  std::vector<int32_t> hi(rows * 2); for (auto &v : hi) v = rand() % n;
  bh_memcpy_h2d(ctx, pool, h.data(), h.size()); bh_memcpy_h2d(ctx, idx, hi.data(), hi.size() * 4); bh_sync(ctx);
  uint32_t key[2] = {0, 7};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) bh_cross(ctx, pool, E, n, idx, n, key, mode, out, gebv, nullptr, nullptr, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) bh_cross(ctx, i & 1 ? out : pool, E, n, idx, n, key, mode, i & 1 ? pool : out, gebv, nullptr, nullptr, 1);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("byte layout mode=%d M=%lld C=%lld : %.1f us per launch, %.1f M crosses/s, %.0f GB/s algorithmic\n", mode, (long long)M, (long long)rows,
         ms / reps * 1e3, rows / (ms / reps * 1e-3) / 1e6, 6.0 * M * rows / (ms / reps * 1e-3) / 1e9);
  return 0;
}
