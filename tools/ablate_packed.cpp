// Ablation driver for the packed cross kernel at BASELINE configs[1] size, through the C ABI.
// usage: ab [mode 0|1] [obs 0|1] [gebv 0|1] [M E n]   (tools/run_ablation_packed.sh builds the variants)
// Timing only -- ablated builds give wrong offspring.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "breedhip.h"

int main(int argc, char **argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0, obs = argc > 2 ? atoi(argv[2]) : 1, wg = argc > 3 ? atoi(argv[3]) : 1;
  const int64_t M = argc > 4 ? atoll(argv[4]) : 10000, E = argc > 5 ? atoll(argv[5]) : 16, n = argc > 6 ? atoll(argv[6]) : 256;
  const int rew = argc > 7 ? atoi(argv[7]) : 0;  // 1: with the per-env reward noted in the launch
  const int reps = getenv("AB_REPS") ? atoi(getenv("AB_REPS")) : 50;  // long runs: clock / power sampling from outside
  std::vector<float> r(M, 0.0015f), eff(M, 1.0f);
  for (int64_t m = 0; m < M; m += M / 10) r[m] = 0.5f;
  bh_ctx *ctx;
  if (bh_create(0, M, 1, r.data(), eff.data(), &ctx)) { printf("create: %s\n", bh_last_error()); return 1; }
  const int64_t pitch = bh_row_pitch(ctx), pp = bh_packed_pitch(ctx), rows = E * n;
  uint8_t *bytes, *ob; uint16_t *pool, *out; int32_t *idx; float *gebv;
  bh_alloc(ctx, rows * pitch, (void **)&bytes); bh_alloc(ctx, rows * pitch, (void **)&ob);
  bh_alloc(ctx, rows * pp, (void **)&pool); bh_alloc(ctx, rows * pp, (void **)&out);
  bh_alloc(ctx, rows * 2 * 4, (void **)&idx); bh_alloc(ctx, rows * 4, (void **)&gebv);
  std::vector<uint8_t> h(rows * pitch);
  { uint64_t s = 88172645463325252ull; uint64_t *w = reinterpret_cast<uint64_t *>(h.data());
    for (size_t i = 0; i < h.size() / 8; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w[i] = s & 0x0101010101010101ull; } }
  // AB_POOL=k: all parents among the first k rows of every env's pool (a cache-resident pool: what the loads cost
  // when they do not go to HBM)
  const int npick = getenv("AB_POOL") ? atoi(getenv("AB_POOL")) : (int)n;
  std::vector<int32_t> hi(rows * 2); for (auto &v : hi) v = rand() % npick;
  bh_memcpy_h2d(ctx, bytes, h.data(), h.size()); bh_memcpy_h2d(ctx, idx, hi.data(), hi.size() * 4);
  bh_pack(ctx, bytes, rows, pool); bh_sync(ctx);
  float *rw[2] = {nullptr, nullptr};
  if (rew) { bh_alloc(ctx, E * 64 * 4, (void **)&rw[0]); bh_alloc(ctx, E * 64 * 4, (void **)&rw[1]); }
  uint32_t key[2] = {0, 7};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) bh_cross_packed(ctx, pool, E, n, idx, n, key, mode, out, obs ? ob : nullptr, wg ? gebv : nullptr, rw[i & 1], rw[1 - (i & 1)], rew ? 64 : 1);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i)
    bh_cross_packed(ctx, i & 1 ? out : pool, E, n, idx, n, key, mode, i & 1 ? pool : out, obs ? ob : nullptr, wg ? gebv : nullptr, rw[(i + 1) & 1], rw[i & 1], rew ? 64 : 1);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("mode=%d obs=%d gebv=%d M=%lld C=%lld : %.1f us per launch\n", mode, obs, wg, (long long)M, (long long)rows, ms / reps * 1e3);
  return 0;
}
