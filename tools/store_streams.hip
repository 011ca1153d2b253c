// How fast does the chip write N row streams of 200 KB each (the byte-layout observation of 8192 x 100k markers)
// when W waves write them, each wave its rows one after the other, VPT KB per visit?  Pure stores, no loads.
// usage: store_streams  (prints a table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int VPT>
__global__ __launch_bounds__(256) void k(uint8_t *out, int n_rows, int64_t pitch, int nvec, int seq) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int first = (blockIdx.x * 4 + wave) * seq;
  const uint4 v = make_uint4(lane, wave, blockIdx.x, 1);
  for (int s = 0; s < seq; ++s) {
    const int row = first + s;
    if (row >= n_rows) return;
    uint8_t *dst = out + (int64_t)row * pitch;
    for (int base = 0; base < nvec; base += 64 * VPT) {
#pragma unroll
      for (int u = 0; u < VPT; ++u) {
        int i = base + 64 * u + lane;
        i = i < nvec ? i : nvec - 1;
        *reinterpret_cast<uint4 *>(dst + 16 * (int64_t)i) = v;
      }
    }
  }
}

int main() {
  const int n_rows = 8192, nvec = 12500;
  const int64_t pitch = 200000;
  uint8_t *out;
  hipMalloc(&out, (size_t)n_rows * pitch);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](int vpt, int seq) {
    const int waves = (n_rows + seq - 1) / seq, blocks = (waves + 3) / 4;
    auto launch = [&]() {
      switch (vpt) {
        case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, n_rows, pitch, nvec, seq); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, n_rows, pitch, nvec, seq); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, n_rows, pitch, nvec, seq); break;
        default: hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, out, n_rows, pitch, nvec, seq); break;
      }
    };
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    printf("streams=%5d (seq=%2d) KB/visit=%d : %7.1f us  %5.2f TB/s\n", waves, seq, vpt, ms * 1e3, n_rows * pitch / (ms * 1e-3) / 1e12);
  };
  for (int seq : {1, 2, 4, 8, 16})
    for (int vpt : {1, 2, 4, 8}) run(vpt, seq);
  return 0;
}
