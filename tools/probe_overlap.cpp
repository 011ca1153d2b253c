// Can the fixed chain of the headline launch hide ACROSS launches?    Through the C ABI, no Python on the
// clock: BASELINE configs[1]'s 16 envs x 256 crosses x 10k markers as ONE context and launch per generation, against
// the same crosses as P parts (P contexts, each on its own HIP stream, each its own dependency chain: a part's next
// generation reads only that part's offspring), issued round-robin from one host thread.  Every launch writes bit
// planes, the byte observation, the GEBV and the per-env reward, as a step's does.  Timing only.
// build: hipcc -O2 -std=c++17 -Iinclude -o tools/_ab/probe_overlap tools/probe_overlap.cpp -Lbreedgym_amd -lbreedhip -Wl,-rpath,'$ORIGIN/../../breedgym_amd'
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "breedhip.h"

struct Part {
  bh_ctx *ctx; hipStream_t s; int64_t E, n;
  uint8_t *bytes, *ob[2]; uint16_t *blk[2]; int32_t *idx; float *gebv; float *rw[2]; int cur = 0;
  uint32_t key[2] = {0, 7};
  int make(int64_t M, int64_t E_, int64_t n_, bool own_stream, const std::vector<float> &r, const std::vector<float> &eff) {
    E = E_; n = n_;
    if (bh_create(0, M, 1, r.data(), eff.data(), &ctx)) { printf("create: %s\n", bh_last_error()); return 1; }
    s = nullptr;
    if (own_stream) { hipStreamCreateWithFlags(&s, hipStreamNonBlocking); bh_set_stream(ctx, s); }
    const int64_t pitch = bh_row_pitch(ctx), pp = bh_packed_pitch(ctx), rows = E * n;
    bh_alloc(ctx, rows * pitch, (void **)&bytes);
    for (int k = 0; k < 2; ++k) {  // two recycled observation blocks and two state blocks, as the env's ring has
      bh_alloc(ctx, rows * pitch, (void **)&ob[k]); bh_alloc(ctx, rows * pp, (void **)&blk[k]);
      bh_alloc(ctx, E * 64 * 4, (void **)&rw[k]);
    }
    bh_alloc(ctx, rows * 2 * 4, (void **)&idx); bh_alloc(ctx, rows * 4, (void **)&gebv);
    std::vector<uint8_t> h(rows * pitch);
    uint64_t x = 88172645463325252ull ^ (uint64_t)E; uint64_t *w = reinterpret_cast<uint64_t *>(h.data());
    for (size_t i = 0; i < h.size() / 8; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w[i] = x & 0x0101010101010101ull; }
    std::vector<int32_t> hi(rows * 2); for (auto &v : hi) v = rand() % (int)n;
    bh_memcpy_h2d(ctx, bytes, h.data(), h.size()); bh_memcpy_h2d(ctx, idx, hi.data(), hi.size() * 4);
    bh_pack(ctx, bytes, rows, blk[0]);
    return bh_sync(ctx);
  }
  int step() {  // the offspring of this generation are the parents of the next
    int rc = bh_cross_packed(ctx, blk[cur], E, n, idx, n, key, BH_RNG_SHARED_ENVS, blk[1 - cur], ob[cur], gebv, rw[cur], rw[1 - cur], 64);
    cur = 1 - cur;
    return rc;
  }
};

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 2000;
  const int max_parts = argc > 2 ? atoi(argv[2]) : 4;  // 1: only the single launch per generation (timing variants of the library)
  const int64_t M = 10000, E = argc > 3 ? atoll(argv[3]) : 16, n = 256;  // E: envs of the whole batch
  std::vector<float> r(M, 0.0015f), eff(M, 1.0f);
  for (int64_t m = 0; m < M; m += M / 10) r[m] = 0.5f;
  for (int P : {1, 2, 4, 1}) {
    if (P > max_parts) continue;
    std::vector<Part> parts(P);
    for (auto &p : parts) if (p.make(M, E / P, n, P > 1, r, eff)) { printf("setup: %s\n", bh_last_error()); return 1; }
    for (int i = 0; i < 3000; ++i) {  // clocks settled, masks made ahead
      for (auto &p : parts) if (p.step()) { printf("step: %s\n", bh_last_error()); return 1; }
      if (i % 64 == 63) hipDeviceSynchronize();
    }
    hipDeviceSynchronize();
    const double t0 = now();
    for (int i = 0; i < reps; ++i) for (auto &p : parts) p.step();
    const double t1 = now();
    hipDeviceSynchronize();
    const double t2 = now();
    printf("%d part(s) of %2lld envs x %lld: %6.2f us per generation of all envs (host issue %5.2f), %6.1f M crosses/s\n", P,
           (long long)(E / P), (long long)n, (t2 - t0) / reps * 1e6, (t1 - t0) / reps * 1e6, E * n * reps / (t2 - t0) / 1e6);
    if (P > 1) {  // one part alone: what a launch of 1/P of the crosses takes
      hipDeviceSynchronize();
      const double a0 = now();
      for (int i = 0; i < reps; ++i) parts[0].step();
      hipDeviceSynchronize();
      printf("    one such part alone: %6.2f us per launch\n", (now() - a0) / reps * 1e6);
    }
    for (auto &p : parts) bh_destroy(p.ctx);
  }
  return 0;
}
