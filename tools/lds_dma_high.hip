// Does LDS-DMA (global_load_lds_dword / _dwordx4, LDS base in M0) reach LDS addresses beyond 64 KB on gfx950?
// One workgroup, 100 KB of dynamic LDS: every probe zeroes the LDS, DMAs 256 B (dword) or 1 KB (dwordx4) of a known
// pattern to LDS offset OFF, then dumps where the pattern actually is.   hipcc --offload-arch=gfx950 -O2 ... && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_p;
typedef const __attribute__((address_space(1))) void *glb_p;
template <int SIZE>
__global__ void probe(const uint32_t *src, int off, int lds_bytes, int *found_at, uint32_t *first_words) {
  extern __shared__ __align__(16) uint8_t lds[];
  const int lane = threadIdx.x;
  for (int i = lane * 4; i < lds_bytes; i += 64 * 4) *reinterpret_cast<uint32_t *>(lds + i) = 0;
  __syncthreads();
  if constexpr (SIZE == 4)
    __builtin_amdgcn_global_load_lds((glb_p)(reinterpret_cast<const char *>(src) + 4 * lane), (lds_p)(lds + off), 4, 0, 0);
  else
    __builtin_amdgcn_global_load_lds((glb_p)(reinterpret_cast<const char *>(src) + 16 * lane), (lds_p)(lds + off), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  if (lane == 0) {
    int at = -1;
    for (int i = 0; i < lds_bytes; i += 4)
      if (*reinterpret_cast<uint32_t *>(lds + i) == 0xA5000000u) { at = i; break; }
    *found_at = at;
    for (int k = 0; k < 4; ++k) first_words[k] = *reinterpret_cast<uint32_t *>(lds + off + 4 * k);
  }
}
int main() {
  const int LDS = 100 * 1024;
  std::vector<uint32_t> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0xA5000000u + i;
  uint32_t *src, *fw; int *found;
  hipMalloc(&src, 4096); hipMalloc(&fw, 16); hipMalloc(&found, 4);
  hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)probe<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipFuncSetAttribute((const void *)probe<16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  const int offs[] = {1024, 32768, 61440, 65536 - 256, 65536, 65536 + 1024, 70000 / 16 * 16, 81920, 98304};
  for (int size : {4, 16})
    for (int off : offs) {
      if (size == 4) hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), LDS, 0, src, off, LDS, found, fw);
      else hipLaunchKernelGGL(probe<16>, dim3(1), dim3(64), LDS, 0, src, off, LDS, found, fw);
      int at; uint32_t w[4];
      hipMemcpy(&at, found, 4, hipMemcpyDeviceToHost); hipMemcpy(w, fw, 16, hipMemcpyDeviceToHost);
      printf("%2d B per lane -> LDS offset %6d: pattern found at %6d  (words at the target: %08x %08x %08x %08x)  %s\n", size, off, at,
             w[0], w[1], w[2], w[3], at == off ? "ok" : "ELSEWHERE");
    }
  return 0;
}
