// What is the most the chip STORES at the headline launch's exact shape?  (DESIGN.md 7.1 says the launch of BASELINE
// configs[1] is bound by its store path: 92 MB in ~13.5 us.  This is the ceiling that claim is measured against.)
//
// The headline launch: 512 workgroups (two per CU), four row waves each, every wave writes two rows -- per row a 20 KB
// byte-layout observation (16 bytes per lane, write-through stores through a buffer descriptor) and 2.5 KB of bit
// planes (2 bytes per lane, plain stores) -- into one of TWO alternating 82 MB + 10 MB block pairs, launch after
// launch: a 184 MB working set that stays in the 256 MB Infinity Cache.  Here: the same grid, the same stores, the same
// alternation, and NOTHING else -- no loads, no arithmetic, no barrier.  Variants: the store policy (plain / sc1 /
// nt), and workgroups that wait `delay` us before their first store (the real launch stores nothing during its first
// ~5 us: dispatch ramp + indices + first words).
//   hipcc -O3 --offload-arch=gfx950 -o store_ceiling tools/store_ceiling.hip && ./store_ceiling
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int kRows = 4096, kNvec = 1250, kNvp = 1256;  // 16 x 256 offspring, 10k markers: 1250 16-byte vectors, 157 x 8 words
constexpr int64_t kPitch = 20000, kPPitch = 2512;

// ALIGNED: the lanes of an instruction shifted along the row so that every instruction's KB (observation) and 128 bytes
// (bit planes) start on a 128-byte line -- rows are 20,000 and 2,512 bytes apart, so as the stream has it (lane l: vector
// 64 u + l) an observation store touches nine lines and a bit-plane store two.
template <int POLICY, bool ALIGNED = false>  // 0 plain, 16 sc1 (write-through), 2 nt
__global__ __launch_bounds__(512) void stores(uint8_t *obs, uint16_t *state, int spin) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= 4) return;  // the mask waves' place
  if (spin > 0) {
    const uint64_t t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (uint64_t)spin) __builtin_amdgcn_s_sleep(8);
  }
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {0x01000100u + lane, 0x00010001u, 0x01010000u, (uint32_t)blockIdx.x};
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = ((int)blockIdx.x * 4 + wave) * 2 + r;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(obs + (int64_t)row * kPitch), 0, (int)kPitch, 0x00020000);
    uint16_t *sp = state + (int64_t)row * kNvp;
    const int so = ALIGNED ? (int)(((int64_t)row * kPitch & 127) >> 4) : 0;   // vectors from the line's start to the row's
    const int ss = ALIGNED ? (int)(((int64_t)row * kPPitch & 127) >> 1) : 0;  // words
    for (int base = 0; base < kNvec + (so > ss ? so : ss); base += 128) {  // two vectors per lane and trip, like the stream
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        int i = base + 64 * u + lane - so, j = base + 64 * u + lane - ss;
        i = i < 0 ? 0 : i < kNvec ? i : kNvec - 1;
        j = j < 0 ? 0 : j < kNvec ? j : kNvec - 1;
        sp[j] = (uint16_t)j;
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, 16 * i, 0, POLICY);
      }
    }
  }
}

int main() {
  uint8_t *obs[2];
  uint16_t *state[2];
  for (int i = 0; i < 2; ++i) {
    hipMalloc(&obs[i], (size_t)kRows * kPitch);
    hipMalloc(&state[i], (size_t)kRows * kPPitch);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const double bytes = (double)kRows * (kPitch + kPPitch);
  auto run = [&](const char *name, int policy, double delay_us) {
    const int spin = (int)(delay_us * 2100.0);  // s_memtime ticks: shader clocks, ~2.1 per ns under this load (profiles/r04_timeline_c2.txt)
    auto launch = [&](int i) {
      if (policy == 0) hipLaunchKernelGGL(stores<0>, dim3(512), dim3(512), 0, 0, obs[i & 1], state[i & 1], spin);
      else if (policy == 16) hipLaunchKernelGGL(stores<16>, dim3(512), dim3(512), 0, 0, obs[i & 1], state[i & 1], spin);
      else if (policy == 17) hipLaunchKernelGGL((stores<16, true>), dim3(512), dim3(512), 0, 0, obs[i & 1], state[i & 1], spin);
      else hipLaunchKernelGGL(stores<2>, dim3(512), dim3(512), 0, 0, obs[i & 1], state[i & 1], spin);
    };
    for (int i = 0; i < 200; ++i) launch(i);  // clocks settle
    hipDeviceSynchronize();
    const int reps = 2000;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch(i);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-34s first store after %4.1f us: %6.2f us per launch  %5.2f TB/s over the launch", name, delay_us, us, bytes / (us * 1e-6) / 1e12);
    if (delay_us > 0) printf("  %5.2f TB/s over the rest", bytes / ((us - delay_us) * 1e-6) / 1e12);
    printf("\n");
  };
  printf("92.2 MB per launch (4096 rows x (20,000 + 2,512) bytes), two block pairs in alternation, 512 workgroups x 4 storing waves\n");
  for (double d : {0.0, 3.0, 5.0}) {
    run("plain stores", 0, d);
    run("write-through (sc1: the stream's)", 16, d);
    run("write-through, instructions on lines", 17, d);
    run("non-temporal (nt)", 2, d);
  }
  return 0;
}
