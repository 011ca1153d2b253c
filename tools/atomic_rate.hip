// What do scattered global atomics cost on gfx950?  (Design question behind the wanted-only shuffle, docs/NOTEBOOK.md 4e: a
// histogram of 21.5 M Threefry words into 2^16 ... 2^18 bins, and appends of ~2.7 M elements to 8192 lists.)
//   hist   : N non-returning atomicAdd(u32) to `bins` random counters
//   append : N returning atomicAdd(u32) to `bins` random counters, the result used (a store at the slot)
//   lds    : the same histogram in LDS (16-bit counters packed in 2^15 words, a workgroup per 65,535 elements),
//            partial histograms written out, no reduction
// usage: atomic_rate   (prints a table)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint32_t mix(uint32_t x) {  // a cheap hash standing in for the Threefry word
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void hist_k(uint32_t *bins, uint32_t mask, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    atomicAdd(&bins[mix((uint32_t)i) & mask], 1u);
}
__global__ __launch_bounds__(256) void append_k(uint32_t *bins, uint32_t mask, int64_t n, uint2 *out, int64_t cap) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t h = mix((uint32_t)i), b = h & mask;
    const uint32_t slot = atomicAdd(&bins[b], 1u);
    out[((int64_t)b * 1024 + (slot & 1023)) % cap] = make_uint2(h, (uint32_t)i);
  }
}
__global__ __launch_bounds__(1024) void lds_k(uint16_t *partial, int64_t n) {
  extern __shared__ uint32_t h2[];  // 2^15 words = 2^16 16-bit counters
  for (int j = threadIdx.x; j < 32768; j += 1024) h2[j] = 0;
  __syncthreads();
  const int64_t c0 = (int64_t)blockIdx.x * 65535, c1 = c0 + 65535 < n ? c0 + 65535 : n;
  for (int64_t i = c0 + threadIdx.x; i < c1; i += 1024) {
    const uint32_t b = mix((uint32_t)i) >> 16;
    atomicAdd(&h2[b >> 1], 1u << (16 * (b & 1)));
  }
  __syncthreads();
  uint4 *dst = reinterpret_cast<uint4 *>(partial + (int64_t)blockIdx.x * 65536);
  for (int j = threadIdx.x; j < 8192; j += 1024) dst[j] = reinterpret_cast<uint4 *>(h2)[j];
}

int main() {
  const int64_t N = 21467628;
  uint32_t *bins; uint2 *out; uint16_t *partial;
  const int64_t cap = (int64_t)1 << 23;
  hipMalloc(&bins, (size_t)4 << 20); hipMalloc(&out, cap * 8);
  const int nblk = (int)((N + 65534) / 65535);
  hipMalloc(&partial, (size_t)nblk * 65536 * 2);
  hipFuncSetAttribute(reinterpret_cast<const void *>(lds_k), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char *what, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s %8.1f us\n", what, ms / 5 * 1e3);
  };
  char name[128];
  for (int bits : {13, 16, 18, 20}) {
    for (int grid : {2048, 8192}) {
      snprintf(name, sizeof name, "hist   N=21.5M non-returning, 2^%d bins, grid %d", bits, grid);
      timeit(name, [&] { hipLaunchKernelGGL(hist_k, dim3(grid), dim3(256), 0, 0, bins, (1u << bits) - 1, N); });
    }
  }
  for (int64_t n : {(int64_t)336000, (int64_t)2700000, N}) {
    for (int bits : {13, 16}) {
      snprintf(name, sizeof name, "append N=%lld returning + 8-byte store, 2^%d counters", (long long)n, bits);
      timeit(name, [&] { hipLaunchKernelGGL(append_k, dim3(2048), dim3(256), 0, 0, bins, (1u << bits) - 1, n, out, cap); });
    }
  }
  snprintf(name, sizeof name, "lds    N=21.5M 2^16 16-bit LDS counters, %d workgroups, partials written", nblk);
  timeit(name, [&] { hipLaunchKernelGGL(lds_k, dim3(nblk), dim3(1024), 131072, 0, partial, N); });
  return 0;
}
