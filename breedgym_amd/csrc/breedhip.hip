// libbreedhip.so -- host side of the C ABI declared in include/breedhip.h.
// HIP only: there is no CPU fallback in this library; without a gfx950 device every
// compute entry point fails with BH_ERR_NO_DEVICE / BH_ERR_HIP.
#include "breedhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <numeric>
#include <vector>

#include "gebv_bits.h"
#include "gebv_mfma.h"
#include "index_kernels.h"
#include "mutation.h"
#include "packed_kernels.h"
#include "row_kernels.h"
#include "select_kernels.h"
#include "threefry.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define BH_HIP(call)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail(e_ == hipErrorOutOfMemory ? BH_ERR_OOM : BH_ERR_HIP, "%s failed: %s",  \
                  #call, hipGetErrorString(e_));                                         \
  } while (0)

constexpr int kMaxFusedTraits = 8;
constexpr int kTopkChunk = 16384;  // candidates one block sorts in LDS (128 KiB of keys)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

}  // namespace

struct bh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t M = 0, T = 0, pitch = 0;
  int32_t half = 0, nwh = 0, nw = 0, nvec = 0;
  int64_t effPitch = 0;
  uint32_t *d_thrA = nullptr, *d_thrB = nullptr;  // 32-bit thresholds (thr23 << 9)
  uint32_t *d_ctrB = nullptr;                     // second Threefry counter per pair
  uint64_t *d_alwA = nullptr, *d_alwB = nullptr;  // site words of markers with r >= 1, or null
  int64_t nthr = 0;
  double *d_eff = nullptr;     // (T, effPitch)
  double *d_effsum = nullptr;  // (effPitch) sum over traits (block values of the optimal haploid)
  // crossover-mask buffers: the launch that streams generation g from d_Z[cur] also makes the
  // masks of generation g+1 in d_Z[1 - cur] (cross_kernel's two roles)
  uint8_t *d_Z[2] = {nullptr, nullptr};
  int64_t z_cap[2] = {0, 0};  // bytes
  uint64_t *d_S = nullptr;    // site words of the stream-per-lane mask role (mask_stream.h)
  uint16_t *d_pk = nullptr;   // bit planes of byte-layout rows on their way to gebv_bits (gebv_chunks)
  int64_t pk_cap = 0;         // bytes
  int64_t s_cap = 0;          // bytes
  float mutation = 0.0f;      // bh_set_mutation: chromax's mutation probability (0: off, the hot path)
  uint32_t mut_thr = 0;       // (bits >> 9) < mut_thr  <=>  uniform_f32(bits) < mutation
  uint64_t *d_F = nullptr;    // flip words of the keyed gametes (mutation.h)
  int64_t f_cap = 0;          // bytes
  bool prefetch = true;
  int layout = bh::LAYOUT_ORIGINAL;  // bh_set_threefry_layout
  // sharding windows: this context holds envs [env_offset, env_offset + n_envs) of envs_total, or crosses
  // [cross_offset, ..) of crosses_total of one batch, and draws the keys the whole would draw (0 = no window)
  int64_t env_offset = 0, envs_total = 0, cross_offset = 0, crosses_total = 0;
  struct {
    bool valid = false;
    int buf = 0;
    uint32_t key0 = 0, key1 = 0;
    int64_t B = 0;
    bool dh = false;
    int64_t off = 0, total = 0;
  } pf;
  int last_buf = 1;
  // fixed-point marker effects for the int8-MFMA GEBV: one pack per group of <= 10 traits
  std::vector<bh::i32x4 *> d_Bpack;
  std::vector<bh::i32x4 *> d_Bbits;  // the same digits in the operand order of gebv_bits_kernel
  int32_t n_mc = 0;                  // 512-marker macro-chunks per row
  double *d_scale = nullptr;  // (T) 2^-s_t
  uint4 *d_dig16 = nullptr;   // (T, 3, nvec) int16 digit planes of the fixed-point effects (packed_kernels.h)
  int32_t n_chunks = 0;
  int32_t *d_partial = nullptr;
  int64_t partial_cap = 0;  // int32 words
  // top-k scratch
  float *d_cand_s[2] = {nullptr, nullptr};
  int32_t *d_cand_i[2] = {nullptr, nullptr};
  int64_t cand_cap = 0;
  // a founder draw asked for ahead (bh_founders_deferred): rides on the next two-role launch, or gets its own
  bh::FoundersParams fp_pending{};
  bool founders_pending = false;
  // reward chaining (bh_cross*): the buffer the last launch left holding -inf, ready to take atomic maxima
  const float *primed_reward = nullptr;
  int64_t primed_envs = 0;
  bool big_lds_set = false;
  bool select_lds_set = false;
  // device-side error word: host memory the kernels can write (a hand-off wait that ran out, row_kernels.h)
  uint32_t *h_err = nullptr, *d_err = nullptr;
};

namespace {

// waves per cross in the standalone mask kernel: long rows are split over the workgroup (LDS per
// cross is M/4 bytes, and the Threefry trips of one cross are independent)
int waves_for(const bh_ctx *c) { return c->M <= 16384 ? 1 : 4; }
// waves per row in the row kernel: only a few long rows are worth four waves each
int row_waves_for(const bh_ctx *c, int64_t n_rows) { return (c->M > 16384 && n_rows < 2048) ? 4 : 1; }

size_t mask_group_lds(const bh_ctx *c) { return (size_t)(c->nwh + 1) * 32; }  // S[nwh + 1][4] u64

// rows one wave streams at once (the effect vectors are loaded once per trip for all of them)
#ifndef BH_ROWS_PER_WAVE_CROSS  // tuning experiments only
#define BH_ROWS_PER_WAVE_CROSS 2
#endif
constexpr int kRowsPerWaveCross = BH_ROWS_PER_WAVE_CROSS, kRowsPerWavePlain = 4;

template <int WAVES, int TT, int MODE, int R>
int launch_row(const bh_ctx *c, bh::RowParams p) {
  const int per_block = R * 4 / WAVES;
  p.n_blocks = (p.n_rows + per_block - 1) / per_block;
  if (R == 2 && c->M > 16384)
    hipLaunchKernelGGL((bh::row_kernel<WAVES, TT, MODE, R, R == 2>), dim3((unsigned)p.n_blocks), dim3(256), 0,
                       c->stream, p);
  else
    hipLaunchKernelGGL((bh::row_kernel<WAVES, TT, MODE, R, false>), dim3((unsigned)p.n_blocks), dim3(256), 0,
                       c->stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BH_ERR_HIP, "row_kernel launch: %s", hipGetErrorString(e));
  return BH_OK;
}

template <int WAVES, int MODE, int R>
int launch_row_tt(const bh_ctx *c, const bh::RowParams &p, int tt) {
  switch (tt) {
    case 0: return launch_row<WAVES, 0, MODE, R>(c, p);
    case 1: return launch_row<WAVES, 1, MODE, R>(c, p);
    case 2: return launch_row<WAVES, 2, MODE, R>(c, p);
    case 4: return launch_row<WAVES, 4, MODE, R>(c, p);
    default: return launch_row<WAVES, 8, MODE, R>(c, p);
  }
}

template <int MODE>
int launch_row_mode(const bh_ctx *c, const bh::RowParams &p, int tt) {
  if (row_waves_for(c, p.n_rows) == 4) return launch_row_tt<4, MODE, 1>(c, p, tt);
  // few rows: fewer per wave keeps more waves in flight
  if (p.n_rows < 4096) return launch_row_tt<1, MODE, 1>(c, p, tt);
  if (MODE == bh::MODE_CROSS || p.n_rows < 16384) return launch_row_tt<1, MODE, kRowsPerWaveCross>(c, p, tt);
  return launch_row_tt<1, MODE, kRowsPerWavePlain>(c, p, tt);
}

// `waves` = waves per cross in the mask role (1 or 4)
// the key window of a call with B keyed crosses: (offset, total) in crosses; total == B without sharding
struct KeyWindow {
  int64_t off, total;
};

bh::MaskParams mask_params(const bh_ctx *c, int buf, uint32_t key0, uint32_t key1, int64_t B, int waves, bool dh,
                           KeyWindow kw) {
  const int groups = 4 / waves;
  bh::MaskParams m{};
  m.Z = c->d_Z[buf];
  m.thrA = c->d_thrA;
  m.thrB = c->d_thrB;
  m.ctrB = c->d_ctrB;
  m.alwA = c->d_alwA;
  m.alwB = c->d_alwB;
  m.thr_bytes = (uint32_t)(c->nthr * 4);
  m.zpitch = (int64_t)16 * c->nw;
  m.M = (int32_t)c->M;
  m.half = c->half;
  m.nwh = c->nwh;
  m.nw = c->nw;
  m.n_rows = (int32_t)B;
  m.n_blocks = (int32_t)((B + groups - 1) / groups);
  m.group_lds = (int32_t)mask_group_lds(c);
  m.key0 = key0;
  m.key1 = key1;
  m.nkeys = (uint32_t)(dh ? kw.total + 1 : 2 * kw.total + 1);
  m.key_offset = (uint32_t)kw.off;
  m.err = c->d_err;
  m.dh = dh;
  m.layout = c->layout;
  return m;
}

template <typename K>
int allow_big_lds(bh_ctx *c, K kern, size_t lds) {
  if (lds <= 64 * 1024) return BH_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
  if (e != hipSuccess) return fail(BH_ERR_HIP, "hipFuncSetAttribute(dynamic LDS): %s", hipGetErrorString(e));
  return BH_OK;
}

// crossover masks of B crosses keyed off `key` into d_Z[buf], in stream order (not overlapped)
int launch_masks(bh_ctx *c, int buf, uint32_t key0, uint32_t key1, int64_t B, bool dh, KeyWindow kw) {
  const int waves = waves_for(c);
  const bh::MaskParams m = mask_params(c, buf, key0, key1, B, waves, dh, kw);
  const size_t lds = (size_t)m.group_lds * (4 / waves);
  if (waves == 1) {
    hipLaunchKernelGGL(bh::mask_kernel<1>, dim3((unsigned)m.n_blocks), dim3(256), lds, c->stream, m);
  } else {
    if (int rc = allow_big_lds(c, bh::mask_kernel<4>, lds)) return rc;
    hipLaunchKernelGGL(bh::mask_kernel<4>, dim3((unsigned)m.n_blocks), dim3(256), lds, c->stream, m);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BH_ERR_HIP, "mask_kernel launch: %s", hipGetErrorString(e));
  return BH_OK;
}

// one launch: waves 0-3 of every workgroup stream `rp` (generation g, R rows per wave), waves 4-7
// fill `mp` (generation g+1) with MW waves per cross.  Only for rows one wave handles.
// Grid of the two-role launch.  With somewhat more mask units than row workgroups (shared keys at
// 2k x 100k: 2048 against 1024) a grid as long as the mask work leaves a tail of mask-only workgroups
// behind the streaming (+220 us measured); the workgroups of a grid as long as the row work take the
// mask units in turns instead (cross_kernel).  With many more mask units than row workgroups the
// launch is mask-bound whatever is done, and short workgroups balance better over the CUs than
// long-lived ones (2.49 ms against 2.10 ms at 8192 independent 100k-marker crosses): no turns then.
// Only for four mask waves per cross: in the one-wave-per-cross kernels the loop costs the registers
// that keep eight waves per SIMD resident.
#ifndef BH_MASK_TURNS_MAX  // tuning experiments only (1 = never take turns)
#define BH_MASK_TURNS_MAX 4
#endif
constexpr int kMaskTurnsMax = BH_MASK_TURNS_MAX;
inline int two_role_grid(int row_blocks, int mask_blocks) {
  if (mask_blocks <= row_blocks) return row_blocks;
  return mask_blocks <= kMaskTurnsMax * row_blocks ? row_blocks : mask_blocks;
}

template <int TT, int MW, int R>
int launch_cross_tt(bh_ctx *c, bh::RowParams rp, const bh::MaskParams &mp) {
  rp.n_blocks = (rp.n_rows + 4 * R - 1) / (4 * R);
  const size_t lds = (size_t)mp.group_lds * (4 / MW);
  const int grid = MW == 4 ? two_role_grid(rp.n_blocks, mp.n_blocks) : std::max(rp.n_blocks, mp.n_blocks);
  if (R == 2 && c->M > 16384)
    hipLaunchKernelGGL((bh::cross_kernel<TT, MW, R, R == 2>), dim3((unsigned)grid), dim3(512), lds, c->stream, rp, mp);
  else
    hipLaunchKernelGGL((bh::cross_kernel<TT, MW, R, false>), dim3((unsigned)grid), dim3(512), lds, c->stream, rp, mp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BH_ERR_HIP, "cross_kernel launch: %s", hipGetErrorString(e));
  return BH_OK;
}

template <int MW, int R>
int launch_cross_mw(bh_ctx *c, const bh::RowParams &rp, const bh::MaskParams &mp, int tt) {
  switch (tt) {
    case 0: return launch_cross_tt<0, MW, R>(c, rp, mp);
    case 1: return launch_cross_tt<1, MW, R>(c, rp, mp);
    case 2: return launch_cross_tt<2, MW, R>(c, rp, mp);
    case 4: return launch_cross_tt<4, MW, R>(c, rp, mp);
    default: return launch_cross_tt<8, MW, R>(c, rp, mp);
  }
}

// ---- packed (bit-plane) rows: packed_kernels.h ------------------------------------------------
// rows one wave streams at once when there are many (BH_PACKED_R: tuning experiments only)
#ifndef BH_PACKED_R
#define BH_PACKED_R 2
#endif
constexpr int kRowsPerWavePacked = BH_PACKED_R;
#ifndef BH_LONG_R  // rows a wave interleaves in the long-row stream (tuning experiments only)
#define BH_LONG_R 1
#endif
constexpr int kRowsPerWaveLong = BH_LONG_R;

template <int TT, int MW, int R>
int launch_pcross_tt(bh_ctx *c, bh::PackedParams rp, const bh::MaskParams &mp) {
  const int per_wave = R * std::max(1, rp.seq);
  rp.n_blocks = (rp.n_rows + 4 * per_wave - 1) / (4 * per_wave);
  // stream-per-lane mask role: no LDS, its waves share the mask work whatever the grid is
  size_t lds = mp.sites ? 0 : (size_t)mp.group_lds * (4 / MW);
  int grid = mp.sites ? rp.n_blocks
                      : (MW == 4 ? two_role_grid(rp.n_blocks, mp.n_blocks) : std::max(rp.n_blocks, mp.n_blocks));
  bh::FoundersParams fp{};
  if (MW == 4 && c->founders_pending && c->fp_pending.P <= 1024 && (size_t)c->fp_pending.P * 16 <= 48 * 1024) {
    fp = c->fp_pending;  // the draw rides on this launch: E more workgroups
    c->founders_pending = false;
    grid += fp.E;
    lds = std::max(lds, (size_t)fp.P * 16);
  }
  hipLaunchKernelGGL((bh::packed_cross_kernel<TT, MW, R>), dim3((unsigned)grid), dim3(512), lds, c->stream, rp, mp, fp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BH_ERR_HIP, "packed_cross_kernel launch: %s", hipGetErrorString(e));
  if (mp.sites) {  // site words -> crossover masks, one wave per keyed cross
    hipLaunchKernelGGL(bh::mask_finish_kernel, dim3((unsigned)((mp.n_rows + 3) / 4)), dim3(256), 0, c->stream, mp);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(BH_ERR_HIP, "mask_finish_kernel launch: %s", hipGetErrorString(e));
  }
  return BH_OK;
}

// the launch for many long rows: stream role in turns + stream-per-lane mask role, then the site words -> masks
#ifndef BH_LONG_MASK_WAVES  // mask waves per workgroup of four row waves (tuning experiments only)
#define BH_LONG_MASK_WAVES 8
#endif
int launch_plong(bh_ctx *c, bh::PackedParams rp, const bh::MaskParams &mp) {
  const int per_wave = kRowsPerWaveLong * std::max(1, rp.seq);
  rp.n_blocks = (rp.n_rows + 4 * per_wave - 1) / (4 * per_wave);
  hipLaunchKernelGGL((bh::packed_long_kernel<kRowsPerWaveLong, BH_LONG_MASK_WAVES>), dim3((unsigned)rp.n_blocks),
                     dim3(64 * (4 + BH_LONG_MASK_WAVES)), 0, c->stream, rp, mp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BH_ERR_HIP, "packed_long_kernel launch: %s", hipGetErrorString(e));
  // site words -> masks, one wave per keyed cross, in stream order (a stream of its own beside the GEBV pass
  // that follows was measured: 811 against 818 us per step at 8192 x 100k -- both want the same memory system)
  hipLaunchKernelGGL(bh::mask_finish_kernel, dim3((unsigned)((mp.n_rows + 3) / 4)), dim3(256), 0, c->stream, mp);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(BH_ERR_HIP, "mask_finish_kernel launch: %s", hipGetErrorString(e));
  return BH_OK;
}

template <int MW, int R>
int launch_pcross_mw(bh_ctx *c, const bh::PackedParams &rp, const bh::MaskParams &mp, int tt) {
  switch (tt) {
    case 0: return launch_pcross_tt<0, MW, R>(c, rp, mp);
    case 1: return launch_pcross_tt<1, MW, R>(c, rp, mp);
    case 2: return launch_pcross_tt<2, MW, R>(c, rp, mp);
    case 4: return launch_pcross_tt<4, MW, R>(c, rp, mp);
    default: return launch_pcross_tt<8, MW, R>(c, rp, mp);
  }
}

template <int TT, int MODE, int R>
int launch_prow(const bh_ctx *c, bh::PackedParams p) {
  const int per_wave = R * std::max(1, p.seq);
  p.n_blocks = (p.n_rows + 4 * per_wave - 1) / (4 * per_wave);
  hipLaunchKernelGGL((bh::packed_row_kernel<TT, MODE, R>), dim3((unsigned)p.n_blocks), dim3(256), 0, c->stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BH_ERR_HIP, "packed_row_kernel launch: %s", hipGetErrorString(e));
  return BH_OK;
}

template <int MODE, int R>
int launch_prow_r(const bh_ctx *c, const bh::PackedParams &p, int tt) {
  switch (tt) {
    case 0: return launch_prow<0, MODE, R>(c, p);
    case 1: return launch_prow<1, MODE, R>(c, p);
    case 2: return launch_prow<2, MODE, R>(c, p);
    case 4: return launch_prow<4, MODE, R>(c, p);
    default: return launch_prow<8, MODE, R>(c, p);
  }
}

template <int MODE>
int launch_prow_mode(const bh_ctx *c, const bh::PackedParams &p, int tt) {
  // few rows: one per wave keeps more waves in flight
  return p.n_rows < 4096 ? launch_prow_r<MODE, 1>(c, p, tt) : launch_prow_r<MODE, kRowsPerWavePacked>(c, p, tt);
}

// few crosses need masks (they are shared by all envs, or the batch is small): put four waves on
// each so that the mask role is as short as the streaming beside it
int mask_waves_for_runahead(const bh_ctx *c, int64_t B) { return (B * 4 <= 2048 || c->M > 16384) ? 4 : 1; }

// make d_Z[buf] hold at least `bytes`; growing it waits for everything that may still touch it
int ensure_z(bh_ctx *c, int buf, int64_t bytes) {
  if (c->z_cap[buf] >= bytes) return BH_OK;
  BH_HIP(hipStreamSynchronize(c->stream));
  if (c->d_Z[buf]) BH_HIP(hipFree(c->d_Z[buf]));
  c->d_Z[buf] = nullptr;
  c->z_cap[buf] = 0;
  if (c->pf.valid && c->pf.buf == buf) c->pf.valid = false;
  BH_HIP(hipMalloc(&c->d_Z[buf], (size_t)bytes));
  c->z_cap[buf] = bytes;
  return BH_OK;
}

// Long bit-plane rows without GEBV in the stream (it follows as the matrix-core pass): a wave streams
// its rows one after the other, so that about BH_LONG_STREAMS rows are being written at any time
// (packed_stream_role).  0 = the ordinary stream.
#ifndef BH_LONG_STREAMS  // tuning experiments only
#define BH_LONG_STREAMS 4096
#endif
int long_row_turns(const bh_ctx *c, int64_t rows) {
  if (c->M <= 16384 || rows < 2 * BH_LONG_STREAMS) return 0;
  return (int)std::min<int64_t>(16, rows / BH_LONG_STREAMS);
}

// site words of B crosses for mask_stream_role: (B, nwh + 1, 4) u64, zero where never written
int ensure_sites(bh_ctx *c, int64_t B) {
  const int64_t bytes = B * (c->nwh + 1) * 32;
  if (c->s_cap >= bytes) return BH_OK;
  BH_HIP(hipStreamSynchronize(c->stream));
  if (c->d_S) BH_HIP(hipFree(c->d_S));
  c->d_S = nullptr;
  c->s_cap = 0;
  BH_HIP(hipMalloc(&c->d_S, (size_t)bytes));
  BH_HIP(hipMemsetAsync(c->d_S, 0, (size_t)bytes, c->stream));
  c->s_cap = bytes;
  return BH_OK;
}

// mutation (mutation.h): draw the flips of the call's keyed gametes and XOR them into the offspring rows the cross
// launch just wrote.  B keyed crosses; `rows` offspring rows of `n_out` per env.
int mutate_rows(bh_ctx *c, uint32_t key0, uint32_t key1, int64_t nkeys, KeyWindow kw, int64_t B, bool dh, bool shared,
                int64_t rows, int64_t n_out, uint16_t *d_out_packed, uint8_t *d_out_bytes) {
  const int32_t half = c->layout == bh::LAYOUT_PARTITIONABLE ? (int32_t)c->M : (int32_t)((c->M + 1) / 2);
  const int32_t wl = (half + 63) / 64;
  const int64_t fpitch = 2 * (int64_t)wl + 1, streams = dh ? B : 2 * B;
  const int64_t bytes = streams * fpitch * 8;
  if (c->f_cap < bytes) {
    BH_HIP(hipStreamSynchronize(c->stream));
    if (c->d_F) BH_HIP(hipFree(c->d_F));
    c->d_F = nullptr;
    c->f_cap = 0;
    BH_HIP(hipMalloc(&c->d_F, (size_t)bytes));
    c->f_cap = bytes;
  }
  BH_HIP(hipMemsetAsync(c->d_F, 0, (size_t)bytes, c->stream));  // the pad word of every stream
  bh::MutationParams sp{};
  sp.F = c->d_F;
  sp.fpitch = fpitch;
  sp.M = (int32_t)c->M;
  sp.half = half;
  sp.wl = wl;
  sp.n_streams = (int32_t)streams;
  sp.key0 = key0;
  sp.key1 = key1;
  sp.nkeys = (uint32_t)nkeys;
  sp.key_offset = (uint32_t)kw.off;
  sp.dh = dh;
  sp.layout = c->layout;
  sp.thr = c->mut_thr;
  for (int64_t s0 = 0; s0 < streams; s0 += 32768) {  // grid.y holds 65535
    const int64_t ns = std::min<int64_t>(32768, streams - s0);
    bh::MutationParams q = sp;
    q.F = sp.F + s0 * fpitch;
    // stream s0 + y: advance the key index by s0 streams (s0 is even: gamete parity is kept)
    q.key_offset = (uint32_t)(kw.off + (dh ? s0 : s0 / 2));
    hipLaunchKernelGGL(bh::mutation_sites_kernel, dim3((unsigned)std::min(64, (wl + 3) / 4), (unsigned)ns), dim3(256), 0,
                       c->stream, q);
    BH_HIP(hipGetLastError());
  }
  bh::MutationApplyParams ap{};
  ap.F = c->d_F;
  ap.fpitch = fpitch;
  ap.M = (int32_t)c->M;
  ap.half = half;
  ap.wl = wl;
  ap.n_rows = (int32_t)rows;
  ap.n_out = (int32_t)n_out;
  ap.shared = shared;
  ap.dh = dh;
  ap.outp = d_out_packed;
  ap.ppitch = 8 * (int64_t)c->nw;
  ap.outb = d_out_bytes;
  ap.bpitch = c->pitch;
  const int64_t threads = rows * ((c->M + 7) / 8);
  if (threads > 0) {
    if ((threads + 255) / 256 > 0x7FFFFFFF) return fail(BH_ERR_UNSUPPORTED, "mutation: too many rows for one launch");
    hipLaunchKernelGGL(bh::mutation_apply_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c->stream, ap);
    BH_HIP(hipGetLastError());
  }
  return BH_OK;
}

int tt_for(int tcount) { return tcount <= 0 ? 0 : tcount == 1 ? 1 : tcount == 2 ? 2 : tcount <= 4 ? 4 : 8; }

bh::RowParams base_params(const bh_ctx *c) {
  bh::RowParams p{};
  p.eff = c->d_eff;
  p.dig = c->d_dig16;
  p.scale = c->d_scale;
  p.pitch = c->pitch;
  p.effPitch = c->effPitch;
  p.zpitch = (int64_t)16 * c->nw;
  p.nvec = c->nvec;
  p.T = (int32_t)c->T;
  return p;
}

bh::PackedParams packed_params(const bh_ctx *c) {
  bh::PackedParams p{};
  p.eff = c->d_eff;
  p.dig = c->d_dig16;
  p.scale = c->d_scale;
  p.pitch = c->pitch;
  p.effPitch = c->effPitch;
  p.zpitch = (int64_t)16 * c->nw;
  p.nvec = c->nvec;
  p.nvp = 8 * c->nw;
  p.T = (int32_t)c->T;
  return p;
}

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
    p.B = c->d_Bpack[g];
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

// GEBV of bit-plane rows on the matrix cores (gebv_bits.h), all traits.  `d_idx` (or null) picks the
// source row of every output row.
#ifndef BH_GEBV_BITS_WGS  // tuning experiments only: workgroups the K slicing aims at
#define BH_GEBV_BITS_WGS 512
#endif
int gebv_bits(bh_ctx *c, const uint16_t *d_rows, int64_t n_src, const int32_t *d_idx, int64_t rows, float *d_gebv) {
  constexpr int kTraitsPerPack = bh::kGebvCols / bh::kGebvDigits;
#ifndef BH_GEBV_BITS_F  // tuning experiments only: 32-row tiles per wave
#define BH_GEBV_BITS_F 1
#endif
  constexpr int F = BH_GEBV_BITS_F;
  const int64_t row_groups = (rows + 128 * F - 1) / (128 * F), rows_pad = row_groups * 128 * F;
  int64_t slices = std::max<int64_t>(1, std::min<int64_t>((BH_GEBV_BITS_WGS + row_groups - 1) / row_groups, c->n_mc));
  const int64_t per_slice = (c->n_mc + slices - 1) / slices;
  slices = (c->n_mc + per_slice - 1) / per_slice;
  const int64_t need = slices * rows_pad * bh::kGebvCols;
  if (need > c->partial_cap) {
    BH_HIP(hipStreamSynchronize(c->stream));
    if (c->d_partial) BH_HIP(hipFree(c->d_partial));
    c->d_partial = nullptr;
    c->partial_cap = 0;
    BH_HIP(hipMalloc(&c->d_partial, (size_t)need * sizeof(int32_t)));
    c->partial_cap = need;
  }
  for (size_t g = 0; g < c->d_Bbits.size(); ++g) {
    const int64_t t0 = (int64_t)g * kTraitsPerPack, tn = std::min<int64_t>(kTraitsPerPack, c->T - t0);
    const int nt = tn * bh::kGebvDigits > 32 ? 2 : 1;
    bh::GebvBitsParams p{};
    p.rows = reinterpret_cast<const uint8_t *>(d_rows);
    p.idx = d_idx;
    p.B = c->d_Bbits[g];
    p.partial = c->d_partial;
    p.row_bytes = (int64_t)16 * c->nw;
    p.n_src = (int32_t)n_src;
    p.n_rows = (int32_t)rows;
    p.rows_pad = (int32_t)rows_pad;
    p.n_mc = c->n_mc;
    p.mc_per_slice = (int32_t)per_slice;
    const dim3 grid((unsigned)row_groups, (unsigned)slices);
    if (nt == 1)
      hipLaunchKernelGGL((bh::gebv_bits_kernel<1, F>), grid, dim3(256), 0, c->stream, p);
    else
      hipLaunchKernelGGL((bh::gebv_bits_kernel<2, F>), grid, dim3(256), 0, c->stream, p);
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
    f.cols = 32 * nt;
    hipLaunchKernelGGL(bh::gebv_finish_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, c->stream, f);
    BH_HIP(hipGetLastError());
  }
  return BH_OK;
}

// from this many traits on, the int8-MFMA GEBV pass beats GEBV fused into the row stream
#ifndef BH_MFMA_GEBV_MIN_TRAITS  // tuning experiments only
#define BH_MFMA_GEBV_MIN_TRAITS 3
#endif
constexpr int kMfmaGebvMinTraits = BH_MFMA_GEBV_MIN_TRAITS;
bool use_mfma_gebv(const bh_ctx *c, int64_t rows) { return c->T >= kMfmaGebvMinTraits && rows >= 32; }
// bit-plane rows: the matrix-core pass (gebv_bits.h) also for one or two traits once the rows are long --
// at 100k markers it takes a quarter of the f64 stream's time (profiles/r02_gebv_bits.txt)
#ifndef BH_BITS_GEBV_MIN_MARKERS  // tuning experiments only
#define BH_BITS_GEBV_MIN_MARKERS 16385
#endif
bool use_bits_gebv(const bh_ctx *c, int64_t rows) {
  return rows >= 32 && (c->T >= kMfmaGebvMinTraits || c->M >= BH_BITS_GEBV_MIN_MARKERS);
}

// GEBV of rows already in memory, for trait chunks [t_begin, T)
int launch_pack(bh_ctx *c, const bh::PackParams &p) {
  for (int64_t r0 = 0; r0 < p.n_rows; r0 += 65535) {  // grid.y holds 65535
    bh::PackParams q = p;
    q.src = p.src + r0 * p.pitch;
    q.dst = p.dst + r0 * p.nvp;
    q.n_rows = (int32_t)std::min<int64_t>(65535, p.n_rows - r0);
    hipLaunchKernelGGL(bh::pack_kernel, dim3((unsigned)((p.nvp + 2047) / 2048), (unsigned)q.n_rows), dim3(256), 0, c->stream, q);
    BH_HIP(hipGetLastError());
  }
  return BH_OK;
}

#ifndef BH_GEBV_VIA_BITS_MIN  // rows x markers from which byte-layout rows go through bit planes (tuning experiments only)
#define BH_GEBV_VIA_BITS_MIN (1ll << 25)
#endif
int gebv_chunks(bh_ctx *c, const uint8_t *d_pop, int64_t rows, float *d_gebv, int t_begin) {
  // large byte-layout populations: to bit planes first (the read of the byte rows at streaming rate, 1/8 of them
  // written), then the matrix-core pass over those (gebv_bits.h) -- 2048 x 100k x 8: 158 -> 115 us; long rows of one
  // or two traits: instead of the f64 stream
  if (t_begin == 0 && use_bits_gebv(c, rows) && rows * c->M >= BH_GEBV_VIA_BITS_MIN) {
    const int64_t bytes = rows * 16 * c->nw;
    if (c->pk_cap < bytes) {
      BH_HIP(hipStreamSynchronize(c->stream));
      if (c->d_pk) BH_HIP(hipFree(c->d_pk));
      c->d_pk = nullptr;
      c->pk_cap = 0;
      BH_HIP(hipMalloc(&c->d_pk, (size_t)bytes));
      c->pk_cap = bytes;
    }
    bh::PackParams pp{};
    pp.src = d_pop;
    pp.dst = c->d_pk;
    pp.pitch = c->pitch;
    pp.nvec = c->nvec;
    pp.nvp = 8 * c->nw;
    pp.n_rows = (int32_t)rows;
    if (int rc = launch_pack(c, pp)) return rc;
    return gebv_bits(c, c->d_pk, rows, nullptr, rows, d_gebv);
  }
  if (t_begin == 0 && use_mfma_gebv(c, rows)) return gebv_mfma(c, d_pop, rows, d_gebv);
  for (int t0 = t_begin; t0 < c->T; t0 += kMaxFusedTraits) {
    bh::RowParams p = base_params(c);
    p.src = d_pop;
    p.gebv = d_gebv;
    p.n_rows = (int32_t)rows;
    p.t0 = t0;
    p.tcount = (int32_t)std::min<int64_t>(kMaxFusedTraits, c->T - t0);
    p.eff = c->d_eff + (int64_t)t0 * c->effPitch;
    int rc = launch_row_mode<bh::MODE_GEBV>(c, p, tt_for(p.tcount));
    if (rc) return rc;
  }
  return BH_OK;
}

// GEBV of packed rows, trait chunks [t_begin, T), fused f64 path (packed_kernels.h)
int packed_gebv_chunks(bh_ctx *c, const uint16_t *d_rows, int64_t rows, float *d_gebv, int t_begin) {
  for (int t0 = t_begin; t0 < c->T; t0 += kMaxFusedTraits) {
    bh::PackedParams p = packed_params(c);
    p.src = d_rows;
    p.gebv = d_gebv;
    p.n_pool = (int32_t)rows;
    p.n_rows = (int32_t)rows;
    p.t0 = t0;
    p.tcount = (int32_t)std::min<int64_t>(kMaxFusedTraits, c->T - t0);
    p.eff = c->d_eff + (int64_t)t0 * c->effPitch;
    if (int rc = launch_prow_mode<bh::MODE_GATHER>(c, p, tt_for(p.tcount))) return rc;
  }
  return BH_OK;
}

int launch_founders(bh_ctx *c, const bh::FoundersParams &p);

// a deferred founder draw that found no launch to ride on is made now, in stream order
int flush_founders(bh_ctx *c) {
  if (!c->founders_pending) return BH_OK;
  c->founders_pending = false;
  return launch_founders(c, c->fp_pending);
}

int launch_founders(bh_ctx *c, const bh::FoundersParams &p) {
  const size_t lds = (size_t)p.P * 16;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bh::founders_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(BH_ERR_HIP, "hipFuncSetAttribute(founders LDS): %s", hipGetErrorString(e));
  }
  const int threads = (int)std::min<int64_t>(1024, std::max<int64_t>(64, p.P / 2));
  hipLaunchKernelGGL(bh::founders_kernel, dim3((unsigned)p.E), dim3(threads), lds, c->stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BH_ERR_HIP, "founders_kernel launch: %s", hipGetErrorString(e));
  return BH_OK;
}

int check_ctx(const bh_ctx *c) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail(BH_ERR_HIP, "hipSetDevice(%d): %s", c->device, hipGetErrorString(e));
  if (c->h_err) {
    const uint32_t w = *static_cast<volatile uint32_t *>(c->h_err);
    if (w)
      return fail(BH_ERR_HIP, "an earlier launch reported a device-side error (code %u: a mask hand-off wait ran out); "
                              "its crossover masks, and every result since, are not to be trusted", w);
  }
  return BH_OK;
}

}  // namespace

extern "C" {

const char *bh_version(void) { return "breedhip 0.2.0 (gfx950)"; }
const char *bh_last_error(void) { return g_err; }

int bh_device_count(int *count) {
  if (!count) return fail(BH_ERR_BAD_ARG, "null count");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(BH_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return BH_OK;
}

int64_t bh_max_markers(void) {
  // LDS per cross ~= M/2 bytes (site words M/4 + interleaved masks M/4); 2^18 markers = 128 KiB
  return (int64_t)1 << 18;
}

int bh_create(int device, int64_t n_markers, int64_t n_traits, const float *recomb_vec,
              const float *effects, bh_ctx **out) {
  if (!out) return fail(BH_ERR_BAD_ARG, "null out");
  *out = nullptr;
  if (n_markers <= 0 || n_traits < 0 || !recomb_vec || (n_traits > 0 && !effects))
    return fail(BH_ERR_BAD_ARG, "bh_create: need n_markers > 0, n_traits >= 0 and non-null arrays");
  if (n_markers > bh_max_markers())
    return fail(BH_ERR_UNSUPPORTED, "bh_create: %lld markers exceed the LDS-staged limit of %lld",
                (long long)n_markers, (long long)bh_max_markers());
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(BH_ERR_NO_DEVICE, "no HIP device visible (libbreedhip has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(BH_ERR_BAD_ARG, "device %d out of range [0,%d)", device, ndev);
  BH_HIP(hipSetDevice(device));

  bh_ctx *c = new (std::nothrow) bh_ctx();
  if (!c) return fail(BH_ERR_OOM, "host allocation failed");
  c->device = device;
  c->M = n_markers;
  c->T = n_traits;
  c->pitch = round_up(2 * n_markers, 16);
  c->half = (int32_t)((n_markers + 1) / 2);
  c->nwh = (c->half + 63) / 64;
  c->nw = (int32_t)((n_markers + 63) / 64);
  c->nvec = (int32_t)(c->pitch / 16);
  c->effPitch = (int64_t)c->nvec * 8;

  // thresholds for the raw 32-bit compare: uniform(bits) < r  <=>  (bits >> 9) < t23  <=>
  // bits < (t23 << 9); t23 = 2^23 (r >= 1, always) does not fit and becomes an always-set bit.
  const int64_t nthr = (int64_t)(c->nwh + 16) * 64;  // + the trips the kernel prefetches ahead
  c->nthr = nthr;
  std::vector<uint32_t> thrA(nthr, 0u), thrB(nthr, 0u), ctrB(nthr, 0u);
  std::vector<uint64_t> alwA((size_t)c->nwh, 0ull), alwB((size_t)c->nwh, 0ull);
  bool any_always = false;
  for (int64_t j = 0; j < c->half; ++j) {
    const uint32_t ta = bh::threshold23(recomb_vec[j]);
    if (ta >> 23) {
      alwA[(size_t)(j >> 6)] |= 1ull << (j & 63);
      any_always = true;
    } else {
      thrA[j] = ta << 9;
    }
    if (j + c->half < n_markers) {
      ctrB[j] = (uint32_t)(j + c->half);
      const uint32_t tb = bh::threshold23(recomb_vec[j + c->half]);
      if (tb >> 23) {
        alwB[(size_t)(j >> 6)] |= 1ull << (j & 63);
        any_always = true;
      } else {
        thrB[j] = tb << 9;
      }
    }
  }
  std::vector<double> eff((size_t)std::max<int64_t>(1, n_traits * c->effPitch), 0.0);
  for (int64_t t = 0; t < n_traits; ++t)
    for (int64_t m = 0; m < n_markers; ++m) eff[t * c->effPitch + m] = (double)effects[m * n_traits + t];

  auto cleanup = [&](int rc) {
    bh_destroy(c);
    return rc;
  };
#define BH_HIP_C(call)                                                                          \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return cleanup(fail(e_ == hipErrorOutOfMemory ? BH_ERR_OOM : BH_ERR_HIP, "%s failed: %s", \
                          #call, hipGetErrorString(e_)));                                       \
  } while (0)
  BH_HIP_C(hipHostMalloc(reinterpret_cast<void **>(&c->h_err), sizeof(uint32_t), hipHostMallocMapped));
  *c->h_err = 0u;
  BH_HIP_C(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_err), c->h_err, 0));
  BH_HIP_C(hipMalloc(&c->d_thrA, nthr * sizeof(uint32_t)));
  BH_HIP_C(hipMalloc(&c->d_thrB, nthr * sizeof(uint32_t)));
  BH_HIP_C(hipMalloc(&c->d_eff, eff.size() * sizeof(double)));
  BH_HIP_C(hipMemcpy(c->d_thrA, thrA.data(), nthr * sizeof(uint32_t), hipMemcpyHostToDevice));
  BH_HIP_C(hipMemcpy(c->d_thrB, thrB.data(), nthr * sizeof(uint32_t), hipMemcpyHostToDevice));
  BH_HIP_C(hipMalloc(&c->d_ctrB, nthr * sizeof(uint32_t)));
  BH_HIP_C(hipMemcpy(c->d_ctrB, ctrB.data(), nthr * sizeof(uint32_t), hipMemcpyHostToDevice));
  if (any_always) {
    BH_HIP_C(hipMalloc(&c->d_alwA, alwA.size() * sizeof(uint64_t)));
    BH_HIP_C(hipMalloc(&c->d_alwB, alwB.size() * sizeof(uint64_t)));
    BH_HIP_C(hipMemcpy(c->d_alwA, alwA.data(), alwA.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    BH_HIP_C(hipMemcpy(c->d_alwB, alwB.data(), alwB.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  }
  BH_HIP_C(hipMemcpy(c->d_eff, eff.data(), eff.size() * sizeof(double), hipMemcpyHostToDevice));
  {
    std::vector<double> effsum((size_t)c->effPitch, 0.0);
    for (int64_t m = 0; m < n_markers; ++m)
      for (int64_t t = 0; t < n_traits; ++t) effsum[(size_t)m] += (double)effects[m * n_traits + t];
    BH_HIP_C(hipMalloc(&c->d_effsum, effsum.size() * sizeof(double)));
    BH_HIP_C(hipMemcpy(c->d_effsum, effsum.data(), effsum.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  // ---- fixed-point effects for the MFMA GEBV (gebv_mfma.h): 6 balanced base-256 digits of
  //      round(eff * 2^s_t), s_t the largest shift that keeps |.| <= 2^46
  if (n_traits > 0) {
    constexpr int kTraitsPerPack = bh::kGebvCols / bh::kGebvDigits;  // 10
    c->n_chunks = (int32_t)((n_markers + 63) / 64);
    std::vector<double> scale((size_t)n_traits);
    std::vector<int> shift((size_t)n_traits);
    for (int64_t t = 0; t < n_traits; ++t) {
      double emax = 0.0, esum = 0.0;
      for (int64_t m = 0; m < n_markers; ++m) {
        const double a = std::fabs((double)effects[m * n_traits + t]);
        emax = std::max(emax, a);
        esum += a;
      }
      int sft = 0;
      if (emax > 0.0 && std::isfinite(esum)) {
        int e2, s2;
        std::frexp(emax, &e2);        // emax = f * 2^e2, f in [0.5, 1)
        std::frexp(2.0 * esum, &s2);  // the largest |GEBV| any genotype can reach: dosage <= 2
        // every fixed-point effect fits the digits (|.| < 2^46), and the int64 in which the digit sums
        // are recombined holds any row: sum_m 2 |round(e_m 2^sft)| <= 2 esum 2^sft + M < 2^61 + 2^18
        sft = std::min(46 - e2, 61 - s2);
      }
      shift[(size_t)t] = sft;
      scale[(size_t)t] = std::ldexp(1.0, -sft);
    }
    BH_HIP_C(hipMalloc(&c->d_scale, (size_t)n_traits * sizeof(double)));
    BH_HIP_C(hipMemcpy(c->d_scale, scale.data(), (size_t)n_traits * sizeof(double), hipMemcpyHostToDevice));
    {  // the same fixed-point effects as three balanced base-65536 digits, 8 markers (one vector) per 16 bytes
      std::vector<int16_t> dig((size_t)n_traits * c->nvec * 3 * 8, (int16_t)0);
      for (int64_t t = 0; t < n_traits; ++t)
        for (int64_t m = 0; m < n_markers; ++m) {
          long long fix = std::llround(std::ldexp((double)effects[m * n_traits + t], shift[(size_t)t]));
          for (int j = 0; j < 3; ++j) {
            long long d = ((fix % 65536) + 65536) % 65536;
            if (d >= 32768) d -= 65536;
            fix = (fix - d) / 65536;
#if defined(BH_DIG_INTERLEAVED)
            dig[(size_t)(((t * c->nvec + (m >> 3)) * 3 + j) * 8 + (m & 7))] = (int16_t)d;
#else
            dig[(size_t)(((t * 3 + j) * c->nvec + (m >> 3)) * 8 + (m & 7))] = (int16_t)d;
#endif
          }
        }
      BH_HIP_C(hipMalloc(&c->d_dig16, dig.size() * sizeof(int16_t)));
      BH_HIP_C(hipMemcpy(c->d_dig16, dig.data(), dig.size() * sizeof(int16_t), hipMemcpyHostToDevice));
    }
    const int64_t packs = (n_traits + kTraitsPerPack - 1) / kTraitsPerPack;
    std::vector<int8_t> pack((size_t)c->n_chunks * 4096);
    for (int64_t g = 0; g < packs; ++g) {
      std::fill(pack.begin(), pack.end(), (int8_t)0);
      const int64_t t0 = g * kTraitsPerPack, tn = std::min<int64_t>(kTraitsPerPack, n_traits - t0);
      for (int64_t m = 0; m < n_markers; ++m) {
        const int64_t ch = m >> 6, w = m & 63, h = w >> 5, q = (w >> 4) & 1, j = w & 15;
        for (int64_t t = 0; t < tn; ++t) {
          long long fix = std::llround(std::ldexp((double)effects[m * n_traits + t0 + t], shift[(size_t)(t0 + t)]));
          for (int d = 0; d < bh::kGebvDigits; ++d) {
            long long dig = ((fix % 256) + 256) % 256;
            if (dig >= 128) dig -= 256;
            fix = (fix - dig) / 256;
            const int64_t col = t * bh::kGebvDigits + d, nt = col >> 5, r = col & 31, lane = r + 32 * h;
            pack[(size_t)((((ch * 2 + nt) * 2 + q) * 64 + lane) * 16 + j)] = (int8_t)dig;
          }
        }
      }
      bh::i32x4 *dB = nullptr;
      BH_HIP_C(hipMalloc(&dB, pack.size()));
      c->d_Bpack.push_back(dB);
      BH_HIP_C(hipMemcpy(dB, pack.data(), pack.size(), hipMemcpyHostToDevice));
    }
    // the same digits in gebv_bits_kernel's order: [mc][q 4][ntile 2][j 4][lane (col, h)][16 markers]
    // <-> marker 512 mc + 256 h + 64 q + 16 j + i (operand byte 4 (i & 3) + (i >> 2)); a macro-chunk of zeros behind the last one
    // (the staging runs two steps ahead)
    c->n_mc = (int32_t)((n_markers + 511) / 512);
    std::vector<int8_t> bits((size_t)(c->n_mc + 1) * 32768);
    for (int64_t g = 0; g < packs; ++g) {
      std::fill(bits.begin(), bits.end(), (int8_t)0);
      const int64_t t0 = g * kTraitsPerPack, tn = std::min<int64_t>(kTraitsPerPack, n_traits - t0);
      for (int64_t m = 0; m < n_markers; ++m) {
        const int64_t mc = m >> 9, rem = m & 511, h = rem >> 8, q = (rem >> 6) & 3, j = (rem >> 4) & 3, i = rem & 15;
        for (int64_t t = 0; t < tn; ++t) {
          long long fix = std::llround(std::ldexp((double)effects[m * n_traits + t0 + t], shift[(size_t)(t0 + t)]));
          for (int d = 0; d < bh::kGebvDigits; ++d) {
            long long dig = ((fix % 256) + 256) % 256;
            if (dig >= 128) dig -= 256;
            fix = (fix - dig) / 256;
            const int64_t col = t * bh::kGebvDigits + d, nt = col >> 5, lane = (col & 31) + 32 * h;
            // operand byte of local marker i: dword i & 3, byte i >> 2 (dosage16_from_bits)
            bits[(size_t)((((((mc * 4 + q) * 2 + nt) * 4 + j) * 64 + lane) * 16) + 4 * (i & 3) + (i >> 2))] = (int8_t)dig;
          }
        }
      }
      bh::i32x4 *dB = nullptr;
      BH_HIP_C(hipMalloc(&dB, bits.size()));
      c->d_Bbits.push_back(dB);
      BH_HIP_C(hipMemcpy(dB, bits.data(), bits.size(), hipMemcpyHostToDevice));
    }
  }
#undef BH_HIP_C
  *out = c;
  return BH_OK;
}

int bh_destroy(bh_ctx *c) {
  if (!c) return BH_OK;
  (void)hipSetDevice(c->device);
  for (int i = 0; i < 2; ++i)
    if (c->d_Z[i]) (void)hipFree(c->d_Z[i]);
  if (c->h_err) (void)hipHostFree(c->h_err);
  if (c->d_S) (void)hipFree(c->d_S);
  if (c->d_pk) (void)hipFree(c->d_pk);
  if (c->d_F) (void)hipFree(c->d_F);
  if (c->d_thrA) (void)hipFree(c->d_thrA);
  if (c->d_thrB) (void)hipFree(c->d_thrB);
  if (c->d_ctrB) (void)hipFree(c->d_ctrB);
  if (c->d_alwA) (void)hipFree(c->d_alwA);
  for (auto *b : c->d_Bpack)
    if (b) (void)hipFree(b);
  for (auto *b : c->d_Bbits)
    if (b) (void)hipFree(b);
  if (c->d_scale) (void)hipFree(c->d_scale);
  if (c->d_dig16) (void)hipFree(c->d_dig16);
  if (c->d_partial) (void)hipFree(c->d_partial);
  if (c->d_alwB) (void)hipFree(c->d_alwB);
  if (c->d_eff) (void)hipFree(c->d_eff);
  if (c->d_effsum) (void)hipFree(c->d_effsum);
  for (int i = 0; i < 2; ++i) {
    if (c->d_cand_s[i]) (void)hipFree(c->d_cand_s[i]);
    if (c->d_cand_i[i]) (void)hipFree(c->d_cand_i[i]);
  }
  delete c;
  return BH_OK;
}

int bh_set_stream(bh_ctx *c, void *hip_stream) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  c->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return BH_OK;
}

int bh_sync(bh_ctx *c) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  BH_HIP(hipStreamSynchronize(c->stream));
  return check_ctx(c);  // ... and whatever the launches that just finished had to report
}

int64_t bh_row_pitch(const bh_ctx *c) { return c ? c->pitch : 0; }
int64_t bh_n_markers(const bh_ctx *c) { return c ? c->M : 0; }
int64_t bh_n_traits(const bh_ctx *c) { return c ? c->T : 0; }

int bh_alloc(bh_ctx *c, int64_t bytes, void **dptr) {
  if (int rc = check_ctx(c)) return rc;
  if (!dptr || bytes < 0) return fail(BH_ERR_BAD_ARG, "bh_alloc: bad arguments");
  *dptr = nullptr;
  BH_HIP(hipMalloc(dptr, (size_t)std::max<int64_t>(bytes, 16)));
  return BH_OK;
}

int bh_free(bh_ctx *c, void *dptr) {
  if (int rc = check_ctx(c)) return rc;
  if (dptr) BH_HIP(hipFree(dptr));
  return BH_OK;
}

int bh_memcpy_h2d(bh_ctx *c, void *dst, const void *src, int64_t bytes) {
  if (int rc = check_ctx(c)) return rc;
  if (bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(BH_ERR_BAD_ARG, "bh_memcpy_h2d: bad arguments");
  if (bytes) BH_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
  return BH_OK;
}

int bh_memcpy_d2h(bh_ctx *c, void *dst, const void *src, int64_t bytes) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  if (bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(BH_ERR_BAD_ARG, "bh_memcpy_d2h: bad arguments");
  if (bytes) BH_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
  BH_HIP(hipStreamSynchronize(c->stream));
  return BH_OK;
}

int bh_upload_population(bh_ctx *c, const uint8_t *host, int64_t rows, uint8_t *d_dst, int validate) {
  if (int rc = check_ctx(c)) return rc;
  if (rows < 0 || (rows > 0 && (!host || !d_dst))) return fail(BH_ERR_BAD_ARG, "bh_upload_population: bad arguments");
  if (rows == 0) return BH_OK;
  const int64_t w = 2 * c->M;
  if (validate) {
    const int64_t n = rows * w;
    uint8_t bad = 0;
    for (int64_t i = 0; i < n; ++i) bad |= host[i];
    if (bad & 0xFE) return fail(BH_ERR_BAD_GENOTYPE, "genotype bytes must be 0 or 1");
  }
  if (c->pitch != w) BH_HIP(hipMemsetAsync(d_dst, 0, (size_t)(rows * c->pitch), c->stream));
  BH_HIP(hipMemcpy2DAsync(d_dst, (size_t)c->pitch, host, (size_t)w, (size_t)w, (size_t)rows,
                          hipMemcpyHostToDevice, c->stream));
  BH_HIP(hipStreamSynchronize(c->stream));  // the host buffer may be pageable and short-lived
  return BH_OK;
}

int bh_download_population(bh_ctx *c, const uint8_t *d_src, int64_t rows, uint8_t *host) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  if (rows < 0 || (rows > 0 && (!host || !d_src))) return fail(BH_ERR_BAD_ARG, "bh_download_population: bad arguments");
  if (rows == 0) return BH_OK;
  const int64_t w = 2 * c->M;
  BH_HIP(hipMemcpy2DAsync(host, (size_t)w, d_src, (size_t)c->pitch, (size_t)w, (size_t)rows,
                          hipMemcpyDeviceToHost, c->stream));
  BH_HIP(hipStreamSynchronize(c->stream));
  return BH_OK;
}

// bh_cross and bh_cross_packed: `packed` selects the layout of the pool and of d_out_packed;
// d_out_bytes is the byte-layout offspring (required unless packed, optional when packed).
static int cross_impl(bh_ctx *c, const char *who, bool packed, const void *d_pool, int64_t n_envs,
                      int64_t pool_rows, const int32_t *d_idx, int64_t n_out, uint32_t sim_key[2], int rng_mode,
                      uint8_t *d_out_bytes, uint16_t *d_out_packed, float *d_gebv, float *d_reward,
                      float *d_reward_next, int64_t reward_pitch) {
  if (int rc = check_ctx(c)) return rc;
  if (!sim_key) return fail(BH_ERR_BAD_ARG, "%s: null sim_key", who);
  if (n_envs < 0 || n_out < 0 || pool_rows < 0) return fail(BH_ERR_BAD_ARG, "%s: negative size", who);
  if ((d_reward || d_reward_next) && reward_pitch < 1) return fail(BH_ERR_BAD_ARG, "%s: reward_pitch must be >= 1", who);
  if (rng_mode < 0 || rng_mode > (BH_RNG_INDEPENDENT | BH_RNG_DOUBLE_HAPLOID))
    return fail(BH_ERR_BAD_ARG, "%s: unknown rng_mode %d", who, rng_mode);
  const bool dh = (rng_mode & BH_RNG_DOUBLE_HAPLOID) != 0;
  const bool shared = (rng_mode & BH_RNG_INDEPENDENT) == 0;
  const int64_t rows = n_envs * n_out;
  const int64_t B = shared ? n_out : rows;
  // independent keys of a shard: the keys the whole batch would split (bh_set_cross_window, or the env window
  // in crosses); shared keys are the same on every shard anyway
  KeyWindow kw{0, B};
  if (!shared) {
    if (c->crosses_total > 0)
      kw = {c->cross_offset, c->crosses_total};
    else if (c->envs_total > 0)
      kw = {c->env_offset * n_out, c->envs_total * n_out};
    if (kw.off + B > kw.total) return fail(BH_ERR_BAD_ARG, "%s: %lld crosses at offset %lld exceed the window of %lld",
                                           who, (long long)B, (long long)kw.off, (long long)kw.total);
  }
  const int64_t nkeys = dh ? kw.total + 1 : 2 * kw.total + 1;
  if (rows > 0x3FFFFFFF || n_envs * pool_rows > 0x7FFFFFFF || 2 * nkeys > 0xFFFFFFFFll)
    return fail(BH_ERR_UNSUPPORTED, "%s: too many rows for 32-bit indexing", who);
  if (rows > 0) {
    if (!d_pool || !d_idx || (packed ? !d_out_packed : !d_out_bytes))
      return fail(BH_ERR_BAD_ARG, "%s: null device pointer", who);
    if (pool_rows == 0) return fail(BH_ERR_BAD_ARG, "%s: empty parent pool", who);
    if (d_gebv && c->T == 0) return fail(BH_ERR_BAD_ARG, "%s: GEBV requested but the context has no traits", who);
    if (d_reward && !d_gebv) return fail(BH_ERR_BAD_ARG, "%s: the reward needs the GEBV output", who);
    const int64_t row_bytes = packed ? (int64_t)16 * c->nw : c->pitch;
    const uint8_t *pool_lo = static_cast<const uint8_t *>(d_pool), *pool_end = pool_lo + n_envs * pool_rows * row_bytes;
    const uint8_t *out_lo = packed ? reinterpret_cast<const uint8_t *>(d_out_packed) : d_out_bytes;
    const uint8_t *out_end = out_lo + rows * row_bytes;
    if (out_lo < pool_end && pool_lo < out_end) return fail(BH_ERR_BAD_ARG, "%s: the output aliases the pool", who);

    // ---- this generation's masks: made by the previous call's launch, or made now
    const int64_t zbytes = B * 16 * c->nw;
    int buf;
    if (c->pf.valid && c->pf.key0 == sim_key[0] && c->pf.key1 == sim_key[1] && c->pf.B == B && c->pf.dh == dh &&
        c->pf.off == kw.off && c->pf.total == kw.total) {
      buf = c->pf.buf;
    } else {  // not predicted: first call, new seed, new batch size
      buf = 1 - c->last_buf;
      if (int rc = ensure_z(c, buf, zbytes)) return rc;
      if (int rc = launch_masks(c, buf, sim_key[0], sim_key[1], B, dh, kw)) return rc;
    }
    c->pf.valid = false;
    c->last_buf = buf;
  }

  // Simulator.cross: keys = split(key, 2B+1); key <- keys[0]
  const bh::u32x2 nk = bh::split_at(sim_key[0], sim_key[1], (uint32_t)nkeys, 0u, c->layout);

  if (rows > 0) {
    const int buf = c->last_buf;
    // GEBV inside the streaming launch (f64 VALU) for few traits; from 3 traits on the int8-MFMA
    // pass over the byte-layout offspring is faster -- when there are byte-layout offspring
    // (bit-plane rows: the pass over the packed offspring, gebv_bits.h)
    // one or two traits leave the stream only where the long-row stream takes over (many long bit-plane rows
    // next to four-wave mask groups): elsewhere the digit products inside the stream are cheaper than a pass
    const bool long_stream = packed && (!c->prefetch || mask_waves_for_runahead(c, B) == 4) && long_row_turns(c, rows) > 1;
    const bool mfma_after =
        d_gebv && (packed ? (c->T >= kMfmaGebvMinTraits ? use_bits_gebv(c, rows) : long_stream) : use_mfma_gebv(c, rows));
    // mutation flips alleles after the crossover: the GEBV (and the reward) must see the mutated rows, so they
    // come from passes behind mutate_rows
    const bool mutate = c->mut_thr != 0;
    const bool fused = d_gebv && c->T <= kMaxFusedTraits && !mfma_after && !mutate;
    const int tcount = fused ? (int32_t)c->T : 0;
    // run ahead: the next generation's masks need only the advanced key and the batch size, so
    // the same launch makes them (mask role) while it streams this generation (row role)
    const bool run_ahead = c->prefetch && (packed || row_waves_for(c, rows) == 1);
    bh::MaskParams mp{};
    int mw = 1;
    if (run_ahead) {
      if (int rc = ensure_z(c, 1 - buf, B * 16 * c->nw)) return rc;
      mw = mask_waves_for_runahead(c, B);
      mp = mask_params(c, 1 - buf, nk.x, nk.y, B, mw, dh, kw);
    }
    const bool wide = rows >= 4096;  // else one row per wave: more waves in flight
    // the per-env maximum comes out of the same launch when the GEBV does: every row wave notes its value
    // with an atomic max into a buffer that holds -inf -- left so by the launch before (d_reward_next of that
    // call), or set so now
    const bool reward_in_launch = d_reward && fused;
    if (reward_in_launch && !(c->primed_reward == d_reward && c->primed_envs == n_envs)) {
      hipLaunchKernelGGL(bh::prime_reward_kernel, dim3(1), dim3(256), 0, c->stream, d_reward, (int)n_envs, reward_pitch);
      BH_HIP(hipGetLastError());
    }
    int rc;
    if (packed) {
      bh::PackedParams p = packed_params(c);
      p.src = static_cast<const uint16_t *>(d_pool);
      p.outp = d_out_packed;
      p.outb = d_out_bytes;
      p.idx = d_idx;
      p.gebv = d_gebv;
      p.Z = c->d_Z[buf];
      p.n_pool = (int32_t)pool_rows;
      p.n_out = (int32_t)n_out;
      p.n_rows = (int32_t)rows;
      p.rng_shared = shared;
      p.tcount = tcount;
      p.reward = reward_in_launch ? d_reward : nullptr;
      p.reward_next = d_reward_next;
      p.reward_pitch = reward_pitch;
      p.n_envs = (int32_t)n_envs;
      const int turns = (tcount == 0 && (!run_ahead || mw == 4)) ? long_row_turns(c, rows) : 0;
      p.seq = turns;
      if (run_ahead) {
        if (mw == 1)
          rc = wide ? launch_pcross_mw<1, kRowsPerWavePacked>(c, p, mp, tt_for(tcount))
                    : launch_pcross_mw<1, 1>(c, p, mp, tt_for(tcount));
        else if (turns > 1) {
#if !defined(BH_NO_MASK_STREAM)  // tuning experiments only: the LDS mask role next to the long-row stream
          if (int rc2 = ensure_sites(c, B)) return rc2;
          mp.sites = c->d_S;
          mp.spitch = c->nwh + 1;
          rc = launch_plong(c, p, mp);
#else
          rc = launch_pcross_mw<4, kRowsPerWaveLong>(c, p, mp, 0);
#endif
        }
        else
          rc = wide ? launch_pcross_mw<4, kRowsPerWavePacked>(c, p, mp, tt_for(tcount))
                    : launch_pcross_mw<4, 1>(c, p, mp, tt_for(tcount));
      } else if (turns > 1) {
        rc = launch_prow_r<bh::MODE_CROSS, kRowsPerWaveLong>(c, p, 0);
      } else {
        rc = launch_prow_mode<bh::MODE_CROSS>(c, p, tt_for(tcount));
      }
    } else {
      bh::RowParams p = base_params(c);
      p.src = static_cast<const uint8_t *>(d_pool);
      p.out = d_out_bytes;
      p.idx = d_idx;
      p.gebv = d_gebv;
      p.Z = c->d_Z[buf];
      p.n_pool = (int32_t)pool_rows;
      p.n_out = (int32_t)n_out;
      p.n_rows = (int32_t)rows;
      p.rng_shared = shared;
      p.t0 = 0;
      p.tcount = tcount;
      p.reward = reward_in_launch ? d_reward : nullptr;
      p.reward_next = d_reward_next;
      p.reward_pitch = reward_pitch;
      p.n_envs = (int32_t)n_envs;
      if (run_ahead) {
        if (mw == 1)
          rc = wide ? launch_cross_mw<1, kRowsPerWaveCross>(c, p, mp, tt_for(tcount))
                    : launch_cross_mw<1, 1>(c, p, mp, tt_for(tcount));
        else
          rc = wide ? launch_cross_mw<4, kRowsPerWaveCross>(c, p, mp, tt_for(tcount))
                    : launch_cross_mw<4, 1>(c, p, mp, tt_for(tcount));
      } else {
        rc = launch_row_mode<bh::MODE_CROSS>(c, p, tt_for(tcount));
      }
    }
    if (rc) return rc;
    if (int rc2 = flush_founders(c)) return rc2;  // a deferred founder draw this kind of launch could not carry
    if (run_ahead) {
      c->pf.valid = true;
      c->pf.buf = 1 - buf;
      c->pf.key0 = nk.x;
      c->pf.key1 = nk.y;
      c->pf.B = B;
      c->pf.dh = dh;
      c->pf.off = kw.off;
      c->pf.total = kw.total;
    }
    if (mutate)
      if (int rc2 = mutate_rows(c, sim_key[0], sim_key[1], nkeys, kw, B, dh, shared, rows, n_out, d_out_packed, d_out_bytes))
        return rc2;
    if (d_gebv && !fused) {
      if (packed)
        rc = mfma_after ? gebv_bits(c, d_out_packed, rows, nullptr, rows, d_gebv)
                        : packed_gebv_chunks(c, d_out_packed, rows, d_gebv, 0);
      else
        rc = gebv_chunks(c, d_out_bytes, rows, d_gebv, 0);
      if (rc) return rc;
    }
    // the reward of vec_env.py:97 where the GEBV came from a pass of its own: a reduction kernel behind it
    if (d_reward && !reward_in_launch) {
      hipLaunchKernelGGL(bh::reward_max_kernel, dim3((unsigned)n_envs), dim3(256), 0, c->stream, d_gebv, n_out * c->T,
                         d_reward, reward_pitch);
      BH_HIP(hipGetLastError());
    }
    c->primed_reward = d_reward_next;
    c->primed_envs = d_reward_next ? n_envs : 0;
  } else if (d_reward_next && n_envs > 0) {
    hipLaunchKernelGGL(bh::prime_reward_kernel, dim3(1), dim3(256), 0, c->stream, d_reward_next, (int)n_envs, reward_pitch);
    BH_HIP(hipGetLastError());
    c->primed_reward = d_reward_next;
    c->primed_envs = n_envs;
  }
  sim_key[0] = nk.x;
  sim_key[1] = nk.y;
  return BH_OK;
}

int bh_cross(bh_ctx *c, const uint8_t *d_pool, int64_t n_envs, int64_t pool_rows, const int32_t *d_idx,
             int64_t n_out, uint32_t sim_key[2], int rng_mode, uint8_t *d_out, float *d_gebv, float *d_reward,
             float *d_reward_next, int64_t reward_pitch) {
  return cross_impl(c, "bh_cross", false, d_pool, n_envs, pool_rows, d_idx, n_out, sim_key, rng_mode, d_out, nullptr,
                    d_gebv, d_reward, d_reward_next, reward_pitch);
}

int bh_cross_packed(bh_ctx *c, const uint16_t *d_pool, int64_t n_envs, int64_t pool_rows, const int32_t *d_idx,
                    int64_t n_out, uint32_t sim_key[2], int rng_mode, uint16_t *d_out_packed, uint8_t *d_out_bytes,
                    float *d_gebv, float *d_reward, float *d_reward_next, int64_t reward_pitch) {
  return cross_impl(c, "bh_cross_packed", true, d_pool, n_envs, pool_rows, d_idx, n_out, sim_key, rng_mode,
                    d_out_bytes, d_out_packed, d_gebv, d_reward, d_reward_next, reward_pitch);
}

int64_t bh_packed_pitch(const bh_ctx *c) { return c ? (int64_t)16 * c->nw : 0; }

int bh_pack(bh_ctx *c, const uint8_t *d_rows, int64_t n_rows, uint16_t *d_packed) {
  if (int rc = check_ctx(c)) return rc;
  if (n_rows < 0) return fail(BH_ERR_BAD_ARG, "bh_pack: negative size");
  if (n_rows == 0) return BH_OK;
  if (!d_rows || !d_packed) return fail(BH_ERR_BAD_ARG, "bh_pack: null device pointer");
  if (n_rows > 0x3FFFFFFF) return fail(BH_ERR_UNSUPPORTED, "bh_pack: too many rows");
  bh::PackParams p{};
  p.src = d_rows;
  p.dst = d_packed;
  p.pitch = c->pitch;
  p.nvec = c->nvec;
  p.nvp = 8 * c->nw;
  p.n_rows = (int32_t)n_rows;
  return launch_pack(c, p);
}

int bh_gather_packed(bh_ctx *c, const uint16_t *d_src, int64_t src_rows, const int32_t *d_idx, int64_t n_rows,
                     uint16_t *d_out_packed, uint8_t *d_out_bytes, float *d_gebv) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  if (n_rows < 0 || src_rows < 0) return fail(BH_ERR_BAD_ARG, "bh_gather_packed: negative size");
  if (n_rows == 0) return BH_OK;
  if (!d_src) return fail(BH_ERR_BAD_ARG, "bh_gather_packed: null device pointer");
  if (src_rows == 0) return fail(BH_ERR_BAD_ARG, "bh_gather_packed: empty source");
  if (!d_idx && n_rows > src_rows) return fail(BH_ERR_BAD_ARG, "bh_gather_packed: more rows than the source holds");
  if (n_rows > 0x3FFFFFFF || src_rows > 0x7FFFFFFF) return fail(BH_ERR_UNSUPPORTED, "bh_gather_packed: too many rows");
  if (d_gebv && c->T == 0)
    return fail(BH_ERR_BAD_ARG, "bh_gather_packed: GEBV requested but the context has no traits");
  bh::PackedParams p = packed_params(c);
  p.src = d_src;
  p.outp = d_out_packed;
  p.outb = d_out_bytes;
  p.idx = d_idx;
  p.gebv = d_gebv;
  p.n_pool = (int32_t)src_rows;
  p.n_rows = (int32_t)n_rows;
  const bool writes = d_out_packed || d_out_bytes;
  const bool mfma_after =
      d_gebv && use_bits_gebv(c, n_rows) &&
      (c->T >= kMfmaGebvMinTraits || !writes || (d_out_packed && long_row_turns(c, n_rows) > 1));
  const bool fused = d_gebv && c->T <= kMaxFusedTraits && !mfma_after;
  p.tcount = fused ? (int32_t)c->T : 0;
  if (p.tcount > 0 || d_out_packed || d_out_bytes) {
    const int turns = (p.tcount == 0 && d_out_packed) ? long_row_turns(c, n_rows) : 0;
    p.seq = turns;
    if (int rc = turns > 1 ? launch_prow_r<bh::MODE_GATHER, kRowsPerWaveLong>(c, p, 0)
                           : launch_prow_mode<bh::MODE_GATHER>(c, p, tt_for(p.tcount)))
      return rc;
  }
  if (d_gebv && !fused) {
    // the matrix-core pass straight over the source rows, through the index (gebv_bits.h)
    if (mfma_after) return gebv_bits(c, d_src, src_rows, d_idx, n_rows, d_gebv);
    // few rows or more than eight traits: f64 passes over the packed rows, through the index
    for (int t0 = 0; t0 < c->T; t0 += kMaxFusedTraits) {
      bh::PackedParams q = p;
      q.outp = nullptr;
      q.outb = nullptr;
      q.t0 = t0;
      q.tcount = (int32_t)std::min<int64_t>(kMaxFusedTraits, c->T - t0);
      q.eff = c->d_eff + (int64_t)t0 * c->effPitch;
      if (int rc = launch_prow_mode<bh::MODE_GATHER>(c, q, tt_for(q.tcount))) return rc;
    }
  }
  return BH_OK;
}

int bh_set_mask_prefetch(bh_ctx *c, int enable) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  c->prefetch = enable != 0;
  return BH_OK;
}

int bh_gather(bh_ctx *c, const uint8_t *d_src, int64_t src_rows, const int32_t *d_idx, int64_t n_rows,
              uint8_t *d_out, float *d_gebv) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  if (n_rows < 0 || src_rows < 0) return fail(BH_ERR_BAD_ARG, "bh_gather: negative size");
  if (n_rows == 0) return BH_OK;
  if (!d_src || !d_idx || !d_out) return fail(BH_ERR_BAD_ARG, "bh_gather: null device pointer");
  if (src_rows == 0) return fail(BH_ERR_BAD_ARG, "bh_gather: empty source");
  if (n_rows > 0x3FFFFFFF || src_rows > 0x7FFFFFFF) return fail(BH_ERR_UNSUPPORTED, "bh_gather: too many rows");
  if (d_gebv && c->T == 0) return fail(BH_ERR_BAD_ARG, "bh_gather: GEBV requested but the context has no traits");
  bh::RowParams p = base_params(c);
  p.src = d_src;
  p.out = d_out;
  p.idx = d_idx;
  p.gebv = d_gebv;
  p.n_pool = (int32_t)src_rows;
  p.n_rows = (int32_t)n_rows;
  const bool fused = d_gebv && c->T <= kMaxFusedTraits && !use_mfma_gebv(c, n_rows);
  p.t0 = 0;
  p.tcount = fused ? (int32_t)c->T : 0;
  if (int rc = launch_row_mode<bh::MODE_GATHER>(c, p, tt_for(p.tcount))) return rc;
  if (d_gebv && !fused) return gebv_chunks(c, d_out, n_rows, d_gebv, 0);
  return BH_OK;
}

int bh_gebv(bh_ctx *c, const uint8_t *d_pop, int64_t rows, float *d_gebv) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  if (rows < 0) return fail(BH_ERR_BAD_ARG, "bh_gebv: negative size");
  if (rows == 0 || c->T == 0) return BH_OK;
  if (!d_pop || !d_gebv) return fail(BH_ERR_BAD_ARG, "bh_gebv: null device pointer");
  if (rows > 0x3FFFFFFF) return fail(BH_ERR_UNSUPPORTED, "bh_gebv: too many rows");
  return gebv_chunks(c, d_pop, rows, d_gebv, 0);
}

int bh_optimal_haploid(bh_ctx *c, const uint8_t *d_pop, int64_t n_rows, const int32_t *d_block_start,
                       int64_t n_blocks, uint8_t *d_ohp, float *d_ohv) {
  if (int rc = check_ctx(c)) return rc;
  if (n_rows < 0 || n_blocks < 0) return fail(BH_ERR_BAD_ARG, "bh_optimal_haploid: negative size");
  if (n_rows == 0) return BH_OK;
  if (!d_pop || (!d_ohp && !d_ohv)) return fail(BH_ERR_BAD_ARG, "bh_optimal_haploid: null device pointer");
  if (n_blocks > 0 && !d_block_start) return fail(BH_ERR_BAD_ARG, "bh_optimal_haploid: blocks without offsets");
  if (c->T == 0) return fail(BH_ERR_BAD_ARG, "bh_optimal_haploid: the context has no traits");
  if (n_rows > 0x3FFFFFFF || n_blocks > 0x7FFFFFFF) return fail(BH_ERR_UNSUPPORTED, "bh_optimal_haploid: too large");
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

int bh_reward_max(bh_ctx *c, const float *d_gebv, int64_t n_envs, int64_t per_env, float *d_reward) {
  if (int rc = check_ctx(c)) return rc;
  if (n_envs < 0 || per_env < 0) return fail(BH_ERR_BAD_ARG, "bh_reward_max: negative size");
  if (n_envs == 0) return BH_OK;
  if (!d_gebv || !d_reward) return fail(BH_ERR_BAD_ARG, "bh_reward_max: null device pointer");
  hipLaunchKernelGGL(bh::reward_max_kernel, dim3((unsigned)n_envs), dim3(256), 0, c->stream, d_gebv, per_env,
                     d_reward, (int64_t)1);
  BH_HIP(hipGetLastError());
  return BH_OK;
}

int bh_topk(bh_ctx *c, const float *d_scores, int64_t n_rows, int64_t n, int64_t k, int32_t *d_idx_out) {
  if (int rc = check_ctx(c)) return rc;
  if (n_rows < 0 || n < 0 || k < 0 || k > n) return fail(BH_ERR_BAD_ARG, "bh_topk: need 0 <= k <= n");
  if (n_rows == 0 || k == 0) return BH_OK;
  if (!d_scores || !d_idx_out) return fail(BH_ERR_BAD_ARG, "bh_topk: null device pointer");
  if (n > 0x7FFFFFFF || n_rows > 65535) return fail(BH_ERR_UNSUPPORTED, "bh_topk: shape too large");
  if (k > kTopkChunk) return fail(BH_ERR_UNSUPPORTED, "bh_topk: k > %d not supported", kTopkChunk);

  if (!c->big_lds_set) {
    BH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bh::topk_chunk_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kTopkChunk * 8));
    c->big_lds_set = true;
  }
  if (n > kTopkChunk) {
    // long vectors: radix select of the k-th key, then sort only the k winners in LDS
    bh::TopkSelectParams q{};
    q.scores = d_scores;
    q.out_idx = d_idx_out;
    q.n = (int32_t)n;
    q.k = (int32_t)k;
    q.P = 2;
    while (q.P < k) q.P <<= 1;
    const size_t lds = (size_t)q.P * 8;
    if (lds > 48 * 1024 && !c->select_lds_set) {
      BH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bh::topk_select_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kTopkChunk * 8));
      c->select_lds_set = true;
    }
    hipLaunchKernelGGL(bh::topk_select_kernel, dim3((unsigned)n_rows), dim3(1024), lds, c->stream, q);
    BH_HIP(hipGetLastError());
    return BH_OK;
  }
  // level 0 reads the caller's scores; every level keeps the best kk of each chunk of <= 16384
  // candidates, until one chunk is left.  top-k of a union = top-k of the parts' top-k.
  const float *in_s = d_scores;
  const int32_t *in_i = nullptr;
  int64_t n_in = n;
  int flip = 0;
  for (;;) {
    const int64_t chunks = (n_in + kTopkChunk - 1) / kTopkChunk;
    const int64_t chunk = chunks == 1 ? n_in : kTopkChunk;
    const int64_t kk = std::min<int64_t>(k, chunk);
    int64_t P = 2;
    while (P < chunk) P <<= 1;
    bh::TopkParams p{};
    p.scores = in_s;
    p.in_idx = in_i;
    p.n_in = (int32_t)n_in;
    p.chunk = (int32_t)chunk;
    p.P = (int32_t)P;
    p.kk = (int32_t)kk;
    if (chunks == 1) {
      p.out_idx = d_idx_out;
      p.out_scores = nullptr;
    } else {
      const int64_t need = n_rows * chunks * kk;
      if (need > c->cand_cap) {
        for (int i = 0; i < 2; ++i) {
          if (c->d_cand_s[i]) (void)hipFree(c->d_cand_s[i]);
          if (c->d_cand_i[i]) (void)hipFree(c->d_cand_i[i]);
          c->d_cand_s[i] = nullptr;
          c->d_cand_i[i] = nullptr;
        }
        c->cand_cap = 0;
        for (int i = 0; i < 2; ++i) {
          BH_HIP(hipMalloc(&c->d_cand_s[i], need * sizeof(float)));
          BH_HIP(hipMalloc(&c->d_cand_i[i], need * sizeof(int32_t)));
        }
        c->cand_cap = need;
      }
      p.out_idx = c->d_cand_i[flip];
      p.out_scores = c->d_cand_s[flip];
    }
    const int threads = (int)std::min<int64_t>(1024, std::max<int64_t>(64, P / 2));
    hipLaunchKernelGGL(bh::topk_chunk_kernel, dim3((unsigned)chunks, (unsigned)n_rows), dim3(threads),
                       (size_t)P * 8, c->stream, p);
    BH_HIP(hipGetLastError());
    if (chunks == 1) break;
    in_s = c->d_cand_s[flip];
    in_i = c->d_cand_i[flip];
    n_in = chunks * kk;
    flip ^= 1;
  }
  return BH_OK;
}

int bh_selection_actions(bh_ctx *c, const float *d_scores, int64_t n_envs, int64_t n, int64_t k, int64_t n_crosses,
                         uint32_t env_key[2], int32_t *d_actions) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  if (!env_key || n_envs < 0 || n <= 0) return fail(BH_ERR_BAD_ARG, "bh_selection_actions: bad sizes");
  const int64_t n_pairs = k * (k - 1) / 2;
  if (k < 2 || k > n) return fail(BH_ERR_BAD_ARG, "bh_selection_actions: need 2 <= k <= n");
  if (n_crosses < 1 || n_crosses > n_pairs || n_crosses > n)
    return fail(BH_ERR_BAD_ARG, "bh_selection_actions: need 1 <= n_crosses <= min(k(k-1)/2, n)");
  if (n > 8192 || n_pairs > 8192 || n_envs > 65535)
    return fail(BH_ERR_UNSUPPORTED, "bh_selection_actions: n and k(k-1)/2 must be <= 8192");
  const int64_t e_total = c->envs_total > 0 ? c->envs_total : n_envs, e_off = c->envs_total > 0 ? c->env_offset : 0;
  if (e_off + n_envs > e_total) return fail(BH_ERR_BAD_ARG, "more envs than the env window holds");
  const bh::u32x2 nk = bh::split_at(env_key[0], env_key[1], (uint32_t)(e_total + 1), 0u, c->layout);
  if (n_envs > 0) {
    if (!d_scores || !d_actions) return fail(BH_ERR_BAD_ARG, "bh_selection_actions: null device pointer");
    bh::SelectionParams p{};
    p.scores = d_scores;
    p.actions = d_actions;
    p.key0 = env_key[0];
    p.key1 = env_key[1];
    p.E = (int32_t)n_envs;
    p.n = (int32_t)n;
    p.k = (int32_t)k;
    p.n_crosses = (int32_t)n_crosses;
    p.n_offspring = (int32_t)((n + n_crosses - 1) / n_crosses);
    p.n_pairs = (int32_t)n_pairs;
    p.P_top = 2;
    while (p.P_top < n) p.P_top <<= 1;
    p.P_pairs = 2;
    while (p.P_pairs < n_pairs) p.P_pairs <<= 1;
    p.rounds = (int32_t)std::ceil(3.0 * std::log((double)std::max<int64_t>(1, n_pairs)) / std::log(4294967295.0));
    p.layout = c->layout;
    p.E_total = (int32_t)e_total;
    p.env_offset = (int32_t)e_off;
    const int PK = std::max(p.P_top, p.P_pairs);
    const size_t lds = (size_t)PK * 8 + (size_t)p.P_pairs * 8 + (size_t)k * 4;
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bh::selection_actions_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return fail(BH_ERR_HIP, "hipFuncSetAttribute(selection LDS): %s", hipGetErrorString(e));
    }
    const int threads = (int)std::min<int64_t>(1024, std::max<int64_t>(64, PK / 2));
    hipLaunchKernelGGL(bh::selection_actions_kernel, dim3((unsigned)n_envs), dim3(threads), lds, c->stream, p);
    BH_HIP(hipGetLastError());
  }
  env_key[0] = nk.x;
  env_key[1] = nk.y;
  return BH_OK;
}

static int founders_device_impl(bh_ctx *c, uint32_t env_key[2], int64_t n_envs, int64_t n_germplasm, int64_t n,
                                int32_t *d_out, bool defer) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  if (!env_key || n_envs < 0 || n < 0 || n > n_germplasm || (n_envs * n > 0 && !d_out))
    return fail(BH_ERR_BAD_ARG, "bh_founders_device: need 0 <= n <= n_germplasm");
  if (n_envs > 65535) return fail(BH_ERR_UNSUPPORTED, "bh_founders_device: too many envs");
  const int64_t e_total = c->envs_total > 0 ? c->envs_total : n_envs, e_off = c->envs_total > 0 ? c->env_offset : 0;
  if (e_off + n_envs > e_total) return fail(BH_ERR_BAD_ARG, "more envs than the env window holds");
  const bh::u32x2 nk = bh::split_at(env_key[0], env_key[1], (uint32_t)(e_total + 1), 0u, c->layout);
  if (n_envs * n > 0) {
    if (n_germplasm > 8192) {  // beyond one workgroup's LDS: draw on the host, copy the indices
      std::vector<int32_t> h((size_t)(e_total * n));
      uint32_t k[2] = {env_key[0], env_key[1]};
      if (int rc = bh_founders(k, e_total, n_germplasm, n, c->layout, h.data())) return rc;
      BH_HIP(hipMemcpyAsync(d_out, h.data() + e_off * n, (size_t)(n_envs * n) * sizeof(int32_t), hipMemcpyHostToDevice,
                            c->stream));
      BH_HIP(hipStreamSynchronize(c->stream));
    } else {
      bh::FoundersParams p{};
      p.out = d_out;
      p.key0 = env_key[0];
      p.key1 = env_key[1];
      p.E = (int32_t)n_envs;
      p.N = (int32_t)n_germplasm;
      p.n = (int32_t)n;
      p.P = 2;
      while (p.P < n_germplasm) p.P <<= 1;
      p.rounds = (int32_t)std::ceil(3.0 * std::log((double)std::max<int64_t>(1, n_germplasm)) / std::log(4294967295.0));
      p.layout = c->layout;
      p.E_total = (int32_t)e_total;
      p.env_offset = (int32_t)e_off;
      if (int rc = (defer ? (c->fp_pending = p, c->founders_pending = true, BH_OK) : launch_founders(c, p))) return rc;
    }
  }
  env_key[0] = nk.x;
  env_key[1] = nk.y;
  return BH_OK;
}

int bh_founders_device(bh_ctx *c, uint32_t env_key[2], int64_t n_envs, int64_t n_germplasm, int64_t n,
                       int32_t *d_out) {
  return founders_device_impl(c, env_key, n_envs, n_germplasm, n, d_out, false);
}

int bh_founders_deferred(bh_ctx *c, uint32_t env_key[2], int64_t n_envs, int64_t n_germplasm, int64_t n,
                         int32_t *d_out) {
  return founders_device_impl(c, env_key, n_envs, n_germplasm, n, d_out, true);
}

void bh_key_from_seed(uint64_t seed, uint32_t key[2]) {
  key[0] = 0u;  // jax with x64 disabled drops the high word
  key[1] = (uint32_t)(seed & 0xFFFFFFFFull);
}

static bool layout_ok(int layout) { return layout == bh::LAYOUT_ORIGINAL || layout == bh::LAYOUT_PARTITIONABLE; }

int bh_set_env_window(bh_ctx *c, int64_t env_offset, int64_t envs_total) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  if (env_offset < 0 || envs_total < 0 || (envs_total > 0 && env_offset >= envs_total) || envs_total > 0x3FFFFFFF)
    return fail(BH_ERR_BAD_ARG, "bh_set_env_window: need 0 <= env_offset < envs_total (or 0, 0)");
  c->env_offset = envs_total > 0 ? env_offset : 0;
  c->envs_total = envs_total;
  c->pf.valid = false;
  return BH_OK;
}

int bh_set_cross_window(bh_ctx *c, int64_t cross_offset, int64_t crosses_total) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  if (cross_offset < 0 || crosses_total < 0 || (crosses_total > 0 && cross_offset >= crosses_total) ||
      crosses_total > 0x3FFFFFFF)
    return fail(BH_ERR_BAD_ARG, "bh_set_cross_window: need 0 <= cross_offset < crosses_total (or 0, 0)");
  c->cross_offset = crosses_total > 0 ? cross_offset : 0;
  c->crosses_total = crosses_total;
  c->pf.valid = false;
  return BH_OK;
}

int bh_set_threefry_layout(bh_ctx *c, int layout) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  if (!layout_ok(layout)) return fail(BH_ERR_BAD_ARG, "bh_set_threefry_layout: unknown layout %d", layout);
  if (layout != c->layout) c->pf.valid = false;  // masks made ahead belong to the other bit stream
  c->layout = layout;
  return BH_OK;
}

int bh_set_mutation(bh_ctx *c, float probability) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  if (!(probability >= 0.0f && probability <= 1.0f))
    return fail(BH_ERR_BAD_ARG, "bh_set_mutation: probability must be in [0, 1]");
  c->mutation = probability;
  c->mut_thr = bh::threshold23(probability);
  return BH_OK;
}

int bh_key_split(const uint32_t key[2], int64_t num, int layout, uint32_t *out) {
  if (!key || !out || num < 0 || num > 0x7FFFFFFF || !layout_ok(layout))
    return fail(BH_ERR_BAD_ARG, "bh_key_split: bad arguments");
  for (int64_t k = 0; k < num; ++k) {
    const bh::u32x2 s = bh::split_at(key[0], key[1], (uint32_t)num, (uint32_t)k, layout);
    out[2 * k] = s.x;
    out[2 * k + 1] = s.y;
  }
  return BH_OK;
}

int bh_permutation(const uint32_t key_in[2], int64_t n, int layout, int32_t *out) {
  if (!key_in || n < 0 || (n > 0 && !out) || n > 0x7FFFFFFF || !layout_ok(layout))
    return fail(BH_ERR_BAD_ARG, "bh_permutation: bad arguments");
  // jax `_shuffle`: ceil(3 ln n / ln(2^32 - 1)) rounds of a stable sort by fresh random words.
  // Sorting (random word << 32 | current position) with an ordinary sort IS that stable sort.
  uint32_t key[2] = {key_in[0], key_in[1]};
  const int rounds = (int)std::ceil(3.0 * std::log((double)std::max<int64_t>(1, n)) / std::log(4294967295.0));
  thread_local std::vector<uint64_t> kv;
  thread_local std::vector<int32_t> x, y;
  kv.resize((size_t)n);
  x.resize((size_t)n);
  y.resize((size_t)n);
  std::iota(x.begin(), x.end(), 0);
  const int64_t L = n + (n & 1), h = L / 2;
  for (int rd = 0; rd < rounds; ++rd) {
    const bh::u32x2 nk = bh::split_at(key[0], key[1], 2u, 0u, layout), sub = bh::split_at(key[0], key[1], 2u, 1u, layout);
    key[0] = nk.x;
    key[1] = nk.y;
    if (layout == bh::LAYOUT_PARTITIONABLE) {
      for (int64_t i = 0; i < n; ++i)
        kv[(size_t)i] = ((uint64_t)bh::bits_word_partitionable(sub.x, sub.y, (uint32_t)i) << 32) | (uint64_t)i;
    } else {
      for (int64_t i = 0; i < h; ++i) {  // random_bits(sub, n): odd n is padded with a 0 counter
        const uint32_t c1 = (i + h < n) ? (uint32_t)(i + h) : 0u;
        const bh::u32x2 o = bh::threefry2x32(sub.x, sub.y, (uint32_t)i, c1);
        kv[(size_t)i] = ((uint64_t)o.x << 32) | (uint64_t)i;
        if (i + h < n) kv[(size_t)(i + h)] = ((uint64_t)o.y << 32) | (uint64_t)(i + h);
      }
    }
    std::sort(kv.begin(), kv.end());
    for (int64_t i = 0; i < n; ++i) y[(size_t)i] = x[(size_t)(uint32_t)kv[(size_t)i]];
    x.swap(y);
  }
  std::copy(x.begin(), x.end(), out);
  return BH_OK;
}

int bh_founders(uint32_t env_key[2], int64_t n_envs, int64_t n_germplasm, int64_t n, int layout, int32_t *out) {
  if (!env_key || n_envs < 0 || n < 0 || n > n_germplasm || (n_envs * n > 0 && !out))
    return fail(BH_ERR_BAD_ARG, "bh_founders: need 0 <= n <= n_germplasm");
  std::vector<uint32_t> keys((size_t)(2 * (n_envs + 1)));
  if (int rc = bh_key_split(env_key, n_envs + 1, layout, keys.data())) return rc;
  std::vector<int32_t> perm((size_t)n_germplasm);
  for (int64_t e = 0; e < n_envs; ++e) {
    if (int rc = bh_permutation(&keys[(size_t)(2 * (e + 1))], n_germplasm, layout, perm.data())) return rc;
    std::copy(perm.begin(), perm.begin() + n, out + e * n);
  }
  env_key[0] = keys[0];
  env_key[1] = keys[1];
  return BH_OK;
}

}  // extern "C"
