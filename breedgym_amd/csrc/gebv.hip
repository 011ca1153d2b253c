// libbreedhip.so -- GEBV on the matrix cores (gebv_mfma.h: byte-layout rows; gebv_bits.h: bit planes) and the
// selection indices (bh_optimal_haploid).
#include "ctx.h"
#include "gebv_bits.h"
#include "gebv_bytes.h"
#include "gebv_mfma.h"
#include "index_kernels.h"

using namespace bhh;

namespace bhh {

// GEBV of rows already in memory on the matrix cores (gebv_mfma.h), all traits
int gebv_mfma(bh_ctx *c, const uint8_t *d_pop, int64_t rows, float *d_gebv) {
  constexpr int kTraitsPerPack = bh::kGebvCols / bh::kGebvDigits;
  const int64_t n_tiles = (rows + 31) / 32, rows_pad = n_tiles * 32;
  int64_t slices = std::max<int64_t>(1, std::min<int64_t>((8192 + n_tiles - 1) / n_tiles, c->n_chunks / 6));
  const int64_t per_slice = (c->n_chunks + slices - 1) / slices;
  slices = (c->n_chunks + per_slice - 1) / per_slice;
  const int64_t need = slices * rows_pad * bh::kGebvCols;
  if (need > c->partial_cap) {
    BH_HIP(hipStreamSynchronize(c->stream));
    if (c->d_partial) BH_HIP(hipFree(c->d_partial));
    c->d_partial = nullptr;
    c->partial_cap = 0;
    BH_HIP(hipMalloc(&c->d_partial, (size_t)need * sizeof(int32_t)));
    c->partial_cap = need;
  }
  for (size_t g = 0; g < c->d_Bpack.size(); ++g) {
    const int64_t t0 = (int64_t)g * kTraitsPerPack, tn = std::min<int64_t>(kTraitsPerPack, c->T - t0);
    bh::GebvMfmaParams p{};
    p.pop = d_pop;
    p.B = static_cast<const bh::i32x4 *>(c->d_Bpack[g]);
    p.partial = c->d_partial;
    p.pitch = c->pitch;
    p.rows = (int32_t)rows;
    p.rows_pad = (int32_t)rows_pad;
    p.n_chunks = c->n_chunks;
    p.chunks_per_slice = (int32_t)per_slice;
    hipLaunchKernelGGL(bh::gebv_mfma_kernel, dim3((unsigned)((n_tiles + 3) / 4), (unsigned)slices), dim3(256), 0,
                       c->stream, p);
    BH_HIP(hipGetLastError());
    bh::GebvFinishParams f{};
    f.partial = c->d_partial;
    f.scale = c->d_scale + t0;
    f.gebv = d_gebv;
    f.rows = (int32_t)rows;
    f.rows_pad = (int32_t)rows_pad;
    f.slices = (int32_t)slices;
    f.T = (int32_t)tn;
    f.Tstride = (int32_t)c->T;
    f.t0 = (int32_t)t0;
    f.cols = bh::kGebvCols;
    hipLaunchKernelGGL(bh::gebv_finish_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, c->stream, f);
    BH_HIP(hipGetLastError());
  }
  return BH_OK;
}

// where gebv_bits_kernel / gebv_bytes_kernel leave their sums (gebv_bits.h, GebvOut): grown on demand, zero whenever
// no launch is in flight (whoever completes a cell leaves it so)
static int gebv_out(bh_ctx *c, int64_t rows, int64_t row_groups, int64_t group_rows, int64_t slices, int64_t t0, int64_t tn,
                    float *d_gebv, bh::GebvOut *o) {
  int64_t gsz = 1;
  while (gsz * gsz < slices) ++gsz;  // ~sqrt(slices) slices per group
  const int64_t n_groups = (slices + gsz - 1) / gsz;
  if (gsz > 127 || n_groups > 127) return fail(BH_ERR_UNSUPPORTED, "GEBV pass: too many K slices for the arrival counts");
  const int64_t cells = row_groups * group_rows * bh::kGsumTraits;
  const int64_t need = (n_groups + 1) * cells * (int64_t)sizeof(unsigned long long);
  if (need > c->gsum_cap) {
    if (int rc = ensure_scratch(c, &c->d_gsum, &c->gsum_cap, need)) return rc;
    BH_HIP(hipMemsetAsync(c->d_gsum, 0, (size_t)need, c->stream));
  }
  o->acc2 = static_cast<unsigned long long *>(c->d_gsum);
  o->acc1 = o->acc2 + cells;
  o->group_stride = cells;
  o->scale = c->d_scale + t0;
  o->gebv = d_gebv;
  o->n_rows = (int32_t)rows;
  o->T = (int32_t)tn;
  o->Tstride = (int32_t)c->T;
  o->t0 = (int32_t)t0;
  o->slices = (int32_t)slices;
  o->group_size = (int32_t)gsz;
  o->n_groups = (int32_t)n_groups;
  return BH_OK;
}

// behind every launch of a pass that leaves through the cells: a launch the runtime refused leaves them as they were,
// but an earlier pack of the same call may be half-way -- the cells are zeroed again behind whatever is in flight, so
// that "all zero whenever no launch is in flight" holds for the next call whatever this one returns
static int launched(bh_ctx *c, const char *what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return BH_OK;
  if (c->d_gsum && c->gsum_cap > 0) (void)hipMemsetAsync(c->d_gsum, 0, (size_t)c->gsum_cap, c->stream);
  return fail(BH_ERR_HIP, "%s launch: %s", what, hipGetErrorString(e));
}

// K slices of a pass over `row_groups` row groups: as many as fill `wgs` workgroup slots, in whole eights when there
// are eight or more -- slice s runs on XCD s % 8 (slice_of_block), 12 slices would give four XCDs twice the work of
// the other four (8192 x 100k x 1 from byte rows: 398 us with 12 slices, 293 us with 22)
// -- and then no more of them than it takes to keep the longest slice as short (2048 x 100k: 48 slices are 4 or 5
// macro-chunks long, 40 slices are 5 nearly all: the same time per workgroup, fewer workgroups in each other's way;
// no XCD gets more slices than before).
static int64_t k_slices(int64_t row_groups, int64_t n_mc, int64_t wgs) {
  int64_t s = std::max<int64_t>(1, std::min<int64_t>(wgs / row_groups, n_mc));
  if (s < 8) return s;
  s -= s % 8;
  const int64_t longest = (n_mc + s - 1) / s;
  return (n_mc + longest - 1) / longest;
}

// GEBV of bit-plane rows on the matrix cores (gebv_bits.h), all traits.  `d_idx` (or null) picks the
// source row of every output row.
constexpr int kGebvBitsWgs = 256;  // workgroups of eight waves: one per CU (112 KB of LDS with two column tiles)
constexpr int kGebvBitsWaves = 4;  // two workgroups of four waves per CU (each ends its superstages with a barrier of its own: 29.3 against 31.5 us at 2048 x 100k x 8, 46.9 against 50.5 at 8192 x 100k x 1 with eight-wave workgroups)
// ... and the byte-row pass: as many workgroups as are resident at once (three per CU: 152 VGPRs, 53 KB of LDS), never
// more -- a second round of a few workgroups costs a whole round's time (784 workgroups: 117 us, 640: 91 us at
// 2048 x 100k x 8, profiles/r03_ab_gebv_bytes.txt)
constexpr int kGebvBytesWgs = 768;  // workgroups the K slicing aims at
// (rows, T) values of a pool's rows through an index (JAX retrieval indexing, as the row kernels read it)
__global__ __launch_bounds__(256) void gebv_through_index_kernel(const float *pool, const int32_t *idx, int n_src, int64_t n, int T,
                                                                 float *out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t row = i / T;
  const int t = (int)(i - row * T);
  int s = idx[row];  // (row_kernels.h, norm_index: negative indices count from the end, what is still outside is clamped)
  if (s < 0) s += n_src;
  s = s < 0 ? 0 : (s >= n_src ? n_src - 1 : s);
  out[i] = pool[(int64_t)s * T + t];
}

int gebv_bits(bh_ctx *c, const uint16_t *d_rows, int64_t n_src, const int32_t *d_idx, int64_t rows, float *d_gebv) {
  constexpr int kTraitsPerPack = bh::kGebvCols / bh::kGebvDigits;
  if (d_idx && rows >= 2 * n_src && n_src >= 32) {
    // Many rows picked out of few (a reset's founders: 4 x 2048 out of a 2048-row germplasm at configs[2]): the pass over
    // the POOL's rows -- a quarter of the bytes there -- and the values through the index; a row's GEBV is a function
    // of its bits alone, so nothing changes but the time (67 -> 26 us per reset at configs[2]).
    const int64_t need = n_src * c->T * (int64_t)sizeof(float);
    if (int rc = ensure_scratch(c, &c->d_pool_gebv, &c->pool_gebv_cap, need)) return rc;
    float *pool = static_cast<float *>(c->d_pool_gebv);
    if (int rc = gebv_bits(c, d_rows, n_src, nullptr, n_src, pool)) return rc;
    const int64_t n = rows * c->T;
    hipLaunchKernelGGL(gebv_through_index_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, pool, d_idx, (int)n_src, n,
                       (int)c->T, d_gebv);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(BH_ERR_HIP, "gebv_through_index launch: %s", hipGetErrorString(e));
    return BH_OK;
  }
  for (size_t g = 0; g < c->d_Bbits.size(); ++g) {
    const int64_t t0 = (int64_t)g * kTraitsPerPack, tn = std::min<int64_t>(kTraitsPerPack, c->T - t0);
    const int nt = tn * bh::kGebvDigits > 32 ? 2 : 1;
    // one column tile (<= 5 traits): the LDS-DMA pass; two: round 3's register-staged pass (gebv_bits.h says why)
    const bool regs = nt == 2;
    const int waves = kGebvBitsWaves;
    const int64_t group_rows = regs ? 128 : bh::bits_rows(waves), row_groups = (rows + group_rows - 1) / group_rows;
    // (the register-staged pass: 512 slots measured against 256 / 384 / 640 / 768 / 1024 at 2048 x 100k x 8 -- 23.7 us against
    // 25.8 / 27.0 / 24.2 / 24.2 / 26.0, round 5)
    const int64_t wgs = regs ? 512 : kGebvBitsWgs * (8 / waves);  // eight waves per CU
    // as many slices as there are workgroup slots, in whole eights (slice s runs on XCD s % 8)
    int64_t slices;
    if (regs) {
      slices = k_slices(row_groups, c->n_mc, wgs + row_groups - 1);
    } else {
      slices = std::max<int64_t>(1, std::min<int64_t>((wgs + row_groups - 1) / row_groups, c->n_mc));
      if (slices >= 8) slices -= slices % 8;
    }
    bh::GebvBitsParams p{};
    p.rows = reinterpret_cast<const uint8_t *>(d_rows);
    p.idx = d_idx;
    p.B = static_cast<const bh::i32x4 *>(c->d_Bbits[g]);
    if (int rc = gebv_out(c, rows, row_groups, group_rows, slices, t0, tn, d_gebv, &p.out)) return rc;
    p.row_bytes = (int64_t)16 * c->pw;
    p.n_src = (int32_t)n_src;
    p.n_rows = (int32_t)rows;
    p.n_mc = c->n_mc;
    p.row_groups = (int32_t)row_groups;
    p.slices = (int32_t)slices;
    const dim3 grid(bh::slice_grid(row_groups, slices));
    if (regs) {
      hipLaunchKernelGGL(bh::gebv_bits_regs_kernel<2>, grid, dim3(256), 0, c->stream, p);
    } else {
      constexpr int lds = bh::gebv_bits_lds<1, kGebvBitsWaves>();
      if (!c->bits_lds_set) {
        BH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bh::gebv_bits_kernel<1, kGebvBitsWaves>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        c->bits_lds_set = 1;
      }
      hipLaunchKernelGGL((bh::gebv_bits_kernel<1, kGebvBitsWaves>), grid, dim3(64 * waves), lds, c->stream, p);
    }
    if (int rc = launched(c, "gebv_bits pass")) return rc;
  }
  return BH_OK;
}

// GEBV of byte-layout rows in one pass: planes built in LDS, then gebv_bits' matrix-core pass (gebv_bytes.h)
int gebv_bytes(bh_ctx *c, const uint8_t *d_pop, int64_t rows, float *d_gebv) {
  constexpr int kTraitsPerPack = kGebvCols / kGebvDigits;
  const int64_t row_groups = (rows + 127) / 128;
  // enough workgroups to keep ~16 KB of row loads per wave in flight on every CU: HBM-bound
  // one full round, two thirds or one third of it -- three, two or one workgroup on EVERY CU (slice s runs on XCD
  // s % 8: whole eights of slices, 32 CUs per XCD) -- beats every count in between: 2048 x 100k x 8 takes 79.6 us with
  // 48 slices, 80.8 with 32, 84.0 with 40; where the row groups do not divide a round, the general rule
  int64_t slices = 0;
  for (int64_t wgs = kGebvBytesWgs; wgs > 0 && !slices; wgs -= kGebvBytesWgs / 3)
    if (wgs % row_groups == 0 && (wgs / row_groups) % 8 == 0 && wgs / row_groups <= c->n_mc) slices = wgs / row_groups;
  if (!slices) slices = k_slices(row_groups, c->n_mc, kGebvBytesWgs);
  for (size_t g = 0; g < c->d_Bbits.size(); ++g) {
    const int64_t t0 = (int64_t)g * kTraitsPerPack, tn = std::min<int64_t>(kTraitsPerPack, c->T - t0);
    const int nt = tn * kGebvDigits > 32 ? 2 : 1;
    bh::GebvBytesParams p{};
    p.rows = d_pop;
    p.B = static_cast<const bh::i32x4 *>(c->d_Bbits[g]);
    if (int rc = gebv_out(c, rows, row_groups, 128, slices, t0, tn, d_gebv, &p.out)) return rc;
    p.pitch = c->pitch;
    p.n_rows = (int32_t)rows;
    p.nvec = c->nvec;
    p.n_mc = c->n_mc;
    p.row_groups = (int32_t)row_groups;
    p.slices = (int32_t)slices;
    const dim3 grid(bh::slice_grid(row_groups, slices));
    if (nt == 1)
      hipLaunchKernelGGL((bh::gebv_bytes_kernel<1>), grid, dim3(256), 0, c->stream, p);
    else
      hipLaunchKernelGGL((bh::gebv_bytes_kernel<2>), grid, dim3(256), 0, c->stream, p);
    if (int rc = launched(c, "gebv_bytes pass")) return rc;
  }
  return BH_OK;
}

}  // namespace bhh

extern "C" {

int bh_optimal_haploid(bh_ctx *c, const uint8_t *d_pop, int64_t n_rows, const int32_t *d_block_start,
                       int64_t n_blocks, uint8_t *d_ohp, float *d_ohv) {
  if (int rc = check_ctx(c)) return rc;
  if (n_rows < 0 || n_blocks < 0) return fail(BH_ERR_BAD_ARG, "bh_optimal_haploid: negative size");
  if (n_rows == 0) return BH_OK;
  if (!d_pop || (!d_ohp && !d_ohv)) return fail(BH_ERR_BAD_ARG, "bh_optimal_haploid: null device pointer");
  if (n_blocks > 0 && !d_block_start) return fail(BH_ERR_BAD_ARG, "bh_optimal_haploid: blocks without offsets");
  if (c->T == 0) return fail(BH_ERR_BAD_ARG, "bh_optimal_haploid: the context has no traits");
  if (n_rows > 0x3FFFFFFF || n_blocks > 0x7FFFFFFF) return fail(BH_ERR_UNSUPPORTED, "bh_optimal_haploid: too large");
  // the stream (index_kernels.h, ohp_stream_kernel) where the blocks' cells fit a workgroup's LDS: the haploid as bits, its
  // value by the matrix-core GEBV pass over them
  const int ohp_rows = bh::kOhpRows * 20 * n_blocks <= bh::kOhpLdsBytes ? bh::kOhpRows : 20 * n_blocks <= bh::kOhpLdsBytes ? 1 : 0;
  if (ohp_rows > 0) {
    const int64_t ppitch = (int64_t)16 * c->pw, planes_bytes = d_ohv ? (n_rows * ppitch + 255) / 256 * 256 : 0;
    const int64_t n_blk = (int64_t)8 * c->nvec, blk_bytes = n_blocks > 0 ? n_blk * 2 : 0;
    if (planes_bytes + blk_bytes > 0)
      if (int rc = ensure_scratch(c, &c->d_ohp, &c->ohp_cap, planes_bytes + blk_bytes)) return rc;
    bh::OhpParams p{};
    p.pop = d_pop;
    p.pos8 = c->d_pos8;
    p.effsum = c->d_effsum;
    p.planes = d_ohv ? static_cast<uint16_t *>(c->d_ohp) : nullptr;
    p.blk = n_blocks > 0 ? reinterpret_cast<const uint16_t *>(static_cast<uint8_t *>(c->d_ohp) + planes_bytes) : nullptr;
    p.ohp = d_ohp;
    p.pitch = c->pitch;
    p.M = (int32_t)c->M;
    p.nvec = c->nvec;
    p.nvp = 8 * c->pw;
    p.n_rows = (int32_t)n_rows;
    p.n_blocks = (int32_t)n_blocks;
    if (n_blocks > 0) {
      hipLaunchKernelGGL(bh::ohp_blocks_kernel, dim3((unsigned)((n_blk + 255) / 256)), dim3(256), 0, c->stream, d_block_start,
                         (int)n_blocks, (int)c->M, (int)n_blk, const_cast<uint16_t *>(p.blk));
      BH_HIP(hipGetLastError());
    }
    const size_t lds = (size_t)ohp_rows * 20 * n_blocks;
    const unsigned grid = (unsigned)((n_rows + ohp_rows - 1) / ohp_rows);
    if (ohp_rows == 1)
      hipLaunchKernelGGL(bh::ohp_stream_kernel<1>, dim3(grid), dim3(256), lds, c->stream, p);
    else
      hipLaunchKernelGGL(bh::ohp_stream_kernel<bh::kOhpRows>, dim3(grid), dim3(256), lds, c->stream, p);
    BH_HIP(hipGetLastError());
    if (d_ohv) return gebv_bits(c, p.planes, n_rows, nullptr, n_rows, d_ohv);
    return BH_OK;
  }
  // ... beyond (more than ~3200 blocks): a wave per row, block by block
  for (int t0 = 0; t0 < c->T; t0 += bh::kOhvTraits) {
    bh::OhvParams p{};
    p.pop = d_pop;
    p.eff = c->d_eff;
    p.effsum = c->d_effsum;
    p.block_start = d_block_start;
    p.ohp = d_ohp;
    p.ohv = d_ohv;
    p.pitch = c->pitch;
    p.effPitch = c->effPitch;
    p.M = (int32_t)c->M;
    p.T = (int32_t)c->T;
    p.t0 = t0;
    p.tcount = (int32_t)std::min<int64_t>(bh::kOhvTraits, c->T - t0);
    p.n_rows = (int32_t)n_rows;
    p.n_blocks = (int32_t)n_blocks;
    hipLaunchKernelGGL(bh::optimal_haploid_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, c->stream, p);
    BH_HIP(hipGetLastError());
    if (!d_ohv) break;  // the haploid itself is complete after the first launch
  }
  return BH_OK;
}

}  // extern "C"
