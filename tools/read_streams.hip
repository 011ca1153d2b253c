// How fast does the chip READ R byte rows of 200 KB (bh_gebv's input at 100k markers) when it is read the way the
// matrix-core pass wants it -- a wave owns 32 rows and walks a slice of the marker axis, visiting every row for V KB
// at a time -- against plain linear streams (a wave reads its rows one after the other)?  Pure loads, 16 KB in
// flight per wave, the loaded words xor-ed into one store per wave.
// usage: read_streams  (prints a table)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

// tile walk: workgroup = 4 waves x 32 rows, K slice = [kb0, kb0 + kbs) KB of every row, V KB per visit of a row
template <int V>
__global__ __launch_bounds__(256) void tile_walk(const uint8_t *in, uint32_t *out, int n_rows, int64_t pitch, int row_kb, int kbs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row_groups = n_rows / 128;
  const int rg = blockIdx.x % row_groups, slice = blockIdx.x / row_groups;
  const int row0 = rg * 128 + wave * 32;
  const int kb0 = slice * kbs, kb1 = min(row_kb, kb0 + kbs);
  uint4 acc = make_uint4(0, 0, 0, 0);
  constexpr int RB = 16 / V;  // rows per batch of 16 loads
  for (int kb = kb0; kb < kb1; kb += V) {
    for (int rb = 0; rb < 32; rb += RB) {
      uint4 v[16];
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int u = 0; u < V; ++u) {
          const int k = min(kb + u, row_kb - 1);
          v[i * V + u] = *reinterpret_cast<const uint4 *>(in + (int64_t)(row0 + rb + i) * pitch + 1024 * (int64_t)k + 16 * lane);
        }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc.x ^= v[i].x;
        acc.y ^= v[i].y;
        acc.z ^= v[i].z;
        acc.w ^= v[i].w;
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = acc.x;
}

// the same 4 KB of a row asked for at about the same time, but by four waves: workgroup = 4 waves x the SAME 32 rows,
// wave w takes KB 4 s + w of every row (16 rows per batch)
__global__ __launch_bounds__(256) void tile_coop(const uint8_t *in, uint32_t *out, int n_rows, int64_t pitch, int row_kb, int kbs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row_groups = n_rows / 32;
  const int rg = blockIdx.x % row_groups, slice = blockIdx.x / row_groups;
  const int row0 = rg * 32;
  const int kb0 = slice * kbs, kb1 = min(row_kb, kb0 + kbs);
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int kb = kb0 + wave; kb < kb1; kb += 4) {
    for (int rb = 0; rb < 32; rb += 16) {
      uint4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i)
        v[i] = *reinterpret_cast<const uint4 *>(in + (int64_t)(row0 + rb + i) * pitch + 1024 * (int64_t)kb + 16 * lane);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc.x ^= v[i].x;
        acc.y ^= v[i].y;
        acc.z ^= v[i].z;
        acc.w ^= v[i].w;
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = acc.x;
}

// the walk as a kernel would really make it: batches of 8 loads (8 / V rows x V KB), D batches in flight, the oldest
// consumed while the newest is issued; workgroup = 4 waves x 32 rows, K slice = [kb0, kb0 + kbs) KB
template <int V, int D>
__global__ __launch_bounds__(256) void tile_pipe(const uint8_t *in, uint32_t *out, int n_rows, int64_t pitch, int row_kb, int kbs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row_groups = n_rows / 128;
  const int rg = blockIdx.x % row_groups, slice = blockIdx.x / row_groups;
  const int row0 = rg * 128 + wave * 32;
  const int kb0 = slice * kbs, kb1 = min(row_kb, kb0 + kbs);
  constexpr int RB = 8 / V;        // rows per batch
  constexpr int NB = 32 / RB;      // batches per visit of the 32 rows
  const int n_batches = (kb1 - kb0 + V - 1) / V * NB;
  uint4 v[D][8];
  uint4 acc = make_uint4(0, 0, 0, 0);
  auto issue = [&](int b, uint4(&dst)[8]) {
    const int kb = kb0 + (b / NB) * V, rb = (b % NB) * RB;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int u = 0; u < V; ++u) {
        const int k = min(kb + u, row_kb - 1);
        dst[i * V + u] = b < n_batches ? *reinterpret_cast<const uint4 *>(in + (int64_t)(row0 + rb + i) * pitch + 1024 * (int64_t)k + 16 * lane)
                                       : make_uint4(0, 0, 0, 0);
      }
  };
#pragma unroll
  for (int d = 0; d < D; ++d) issue(d, v[d]);
  for (int b = 0; b < n_batches; b += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc.x ^= v[d][i].x;
        acc.y ^= v[d][i].y;
        acc.z ^= v[d][i].z;
        acc.w ^= v[d][i].w;
      }
      issue(b + d + D, v[d]);
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = acc.x;
}

// linear: a wave reads `seq` whole rows one after the other, 16 KB per trip
__global__ __launch_bounds__(256) void linear(const uint8_t *in, uint32_t *out, int n_rows, int64_t pitch, int row_kb, int seq) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int first = (blockIdx.x * 4 + wave) * seq;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int s = 0; s < seq; ++s) {
    const int row = first + s;
    if (row >= n_rows) break;
    for (int kb = 0; kb < row_kb; kb += 16) {
      uint4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int k = min(kb + u, row_kb - 1);
        v[u] = *reinterpret_cast<const uint4 *>(in + (int64_t)row * pitch + 1024 * (int64_t)k + 16 * lane);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc.x ^= v[i].x;
        acc.y ^= v[i].y;
        acc.z ^= v[i].z;
        acc.w ^= v[i].w;
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = acc.x;
}

int main() {
  const int row_kb = 195;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int64_t pitch : {200000})
  for (int n_rows : {2048, 8192}) {
    uint8_t *in;
    uint32_t *out;
    hipMalloc(&in, (size_t)n_rows * pitch);
    hipMalloc(&out, 1 << 24);
    hipMemset(in, 1, (size_t)n_rows * pitch);
    auto timed = [&](const char *what, int a, int b, auto launch) {
      for (int i = 0; i < 3; ++i) launch();
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      for (int i = 0; i < 10; ++i) launch();
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= 10;
      printf("pitch=%ld rows=%5d %-22s %4d %4d : %7.1f us  %5.2f TB/s\n", (long)pitch, n_rows, what, a, b, ms * 1e3, n_rows * 1024.0 * row_kb / (ms * 1e-3) / 1e12);
      fflush(stdout);
    };
    for (int slices : {12, 24, 49, 98}) {
      const int kbs = (row_kb + slices - 1) / slices;
      const int blocks = n_rows / 128 * ((row_kb + kbs - 1) / kbs);
      if (blocks > 16384) continue;
      timed("tile V=1 slices,wgs", slices, blocks, [&]() { hipLaunchKernelGGL(tile_walk<1>, dim3(blocks), dim3(256), 0, 0, in, out, n_rows, pitch, row_kb, kbs); });
      timed("tile V=2 slices,wgs", slices, blocks, [&]() { hipLaunchKernelGGL(tile_walk<2>, dim3(blocks), dim3(256), 0, 0, in, out, n_rows, pitch, row_kb, kbs); });
      timed("tile V=4 slices,wgs", slices, blocks, [&]() { hipLaunchKernelGGL(tile_walk<4>, dim3(blocks), dim3(256), 0, 0, in, out, n_rows, pitch, row_kb, kbs); });
    }
    for (int kbs : {4, 5, 8, 10, 12, 16}) {  // KB of K per workgroup: 640 / 400 workgroups at 2048 rows with 5 / 8
      const int blocks = n_rows / 128 * ((row_kb + kbs - 1) / kbs);
      timed("pipe V=1 D=2 kbs,wgs", kbs, blocks, [&]() { hipLaunchKernelGGL((tile_pipe<1, 2>), dim3(blocks), dim3(256), 0, 0, in, out, n_rows, pitch, row_kb, kbs); });
      timed("pipe V=2 D=2 kbs,wgs", kbs, blocks, [&]() { hipLaunchKernelGGL((tile_pipe<2, 2>), dim3(blocks), dim3(256), 0, 0, in, out, n_rows, pitch, row_kb, kbs); });
      timed("pipe V=4 D=2 kbs,wgs", kbs, blocks, [&]() { hipLaunchKernelGGL((tile_pipe<4, 2>), dim3(blocks), dim3(256), 0, 0, in, out, n_rows, pitch, row_kb, kbs); });
      timed("pipe V=4 D=3 kbs,wgs", kbs, blocks, [&]() { hipLaunchKernelGGL((tile_pipe<4, 3>), dim3(blocks), dim3(256), 0, 0, in, out, n_rows, pitch, row_kb, kbs); });
    }
    for (int slices : {3, 6, 12, 24}) {
      const int kbs = (row_kb + slices - 1) / slices;
      const int blocks = n_rows / 32 * ((row_kb + kbs - 1) / kbs);
      timed("coop V=1 slices,wgs", slices, blocks, [&]() { hipLaunchKernelGGL(tile_coop, dim3(blocks), dim3(256), 0, 0, in, out, n_rows, pitch, row_kb, kbs); });
    }
    for (int seq : {1, 2, 4}) {
      const int blocks = (n_rows / seq + 3) / 4;
      timed("linear seq,wgs", seq, blocks, [&]() { hipLaunchKernelGGL(linear, dim3(blocks), dim3(256), 0, 0, in, out, n_rows, pitch, row_kb, seq); });
    }
    hipFree(in);
    hipFree(out);
  }
  return 0;
}
