// Timing driver for the GEBV passes through the C ABI: from bit-plane rows (bh_gather_packed with only the GEBV
// output) and from byte-layout rows (bh_gebv).
// usage: driver [rows] [M] [T]     (tools/ab_build.sh --driver tools/ablate_gebv.cpp builds variants)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "breedhip.h"

int main(int argc, char **argv) {
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 2048, M = argc > 2 ? atoll(argv[2]) : 100000, T = argc > 3 ? atoll(argv[3]) : 8;
  const int reps = 50;
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
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) bh_gather_packed(ctx, pool, rows, nullptr, rows, nullptr, nullptr, gebv);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) bh_gather_packed(ctx, pool, rows, nullptr, rows, nullptr, nullptr, gebv);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("rows=%lld M=%lld T=%lld : bit planes %.1f us per call", (long long)rows, (long long)M, (long long)T, ms / reps * 1e3);
  uint8_t *bytes;
  const int64_t pitch = bh_row_pitch(ctx);
  bh_alloc(ctx, rows * pitch, (void **)&bytes);
  { std::vector<uint64_t> hb(rows * pitch / 8); for (auto &w : hb) w = rnd() & 0x0101010101010101ull;
    bh_memcpy_h2d(ctx, bytes, hb.data(), hb.size() * 8); bh_sync(ctx); }
  for (int i = 0; i < 5; ++i) bh_gebv(ctx, bytes, rows, gebv);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) bh_gebv(ctx, bytes, rows, gebv);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf(", byte rows %.1f us per call (%.2f TB/s)\n", ms / reps * 1e3, rows * pitch / (ms / reps * 1e-3) / 1e12);
  return 0;
}
